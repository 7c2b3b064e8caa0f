#!/usr/bin/env python3
"""bench.py -- forward+adjoint mul! pairs/sec on a tall JopBlock (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): the configuration the metric is quoted on -- a 1024x1 tall JopBlock of
diagonal blocks, 256^3 Float32 per block (64 MiB/block; 64 GiB of coefficients + 64 GiB range
vector), synthetic U[0,1) data from the counter-based generator (seeds a=1, m=2, d=3), resident in
HBM before the timed region.  One step = one forward mul!(d, A, m) + one adjoint mul!(m, A', d).
With N GPUs the SAME operator is row-partitioned (1024/N block rows per rank, "strong" scaling);
the adjoint ends with one RCCL all-reduce of the 64 MiB domain vector.

Launch modes for N > 1 (--mode, default auto).  Whatever started us -- a bare `python bench.py --gpus N` or N copies under
torch.distributed.run -- the process that was started NEVER touches the GPU: it is a supervisor that starts the real workers as
child processes, watches their heartbeats and relays the outcome.
  ranks : one worker process per GPU over torch.distributed / RCCL (the deployment model; what the driver's launch line means).
  team  : ONE worker process drives all N GPUs through the C ABI's single-process team (jh_comm_init_all = ncclCommInitAll,
          one context per device, grouped ranged all-reduces) -- needs no N GPU-holding processes and no IPC handles.
  auto  : ranks first; if a rank dies or stalls BEFORE the first collective has completed (process cap, RCCL bootstrap), the
          supervisor stops exactly the PIDs it started and runs a FRESH team-mode child instead, and the line says so
          ("launch_mode", "launch_fallback").  A failure after the first collective is a real failure: non-zero exit, no line.
Workers write heartbeats (phase names) to a file; no heartbeat for BENCH_WATCHDOG_S (120) seconds -- BENCH_WATCHDOG_IMPORT_S
(300) while python / torch are still being paged in and until the first collective has completed (first contact with RCCL) --
stops the job with "rank r stalled in phase p".

Extras (not the metric): --fused-normal (the fused A'A kernel), --lsqr K (K LSQR iterations on b = A x_true), --check (team
mode: replicas bit-identical, adjoint against the fp64 sum of the members' partial sums; --dump FILE saves member 0's result).
BENCH_FORCE_DIST=1 under torch.distributed.run with one process exercises the RCCL path on a one-GPU box.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed on the library stream) and `cpu_baseline` (the CPU oracle, single thread, on a
bounded sample; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
def cpu_share() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup's CPU quota (a GPU box of this pool shows 256 CPUs
    and grants 16: 128 OpenMP threads fighting over a 16-CPU quota made the all-cores baseline swing between 0.17 and 1.96 pairs/s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


_USER_SET_OMP_THREADS = "OMP_NUM_THREADS" in os.environ
CPU_SHARE = cpu_share()                                # (once, at import: OpenMP's pinning narrows the main thread's affinity mask later)
os.environ.setdefault("OMP_NUM_THREADS", str(CPU_SHARE))   # the all-cores CPU baseline uses the CPUs this process is granted, no more
os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # the all-cores CPU baseline must not spin on barriers in a CPU-capped container
os.environ.setdefault("OMP_PROC_BIND", "close")       # ... and its threads stay where they first touched their pages
os.environ.setdefault("OMP_PLACES", "cores")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nblocks", type=int, default=1024, help="global block rows of the tall operator")
    ap.add_argument("--edge", type=int, default=256, help="block is edge^3 Float32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-blocks", type=int, default=16)
    ap.add_argument("--cpu-pairs", type=int, default=4)
    ap.add_argument("--tune", type=str, default="", help="k=v,k=v kernel knobs (fwd_group, fwd_unroll, fwd_wg, adj_unroll, adj_depth, adj_wg, nt; 0 = automatic)")
    ap.add_argument("--fused-normal", action="store_true", help="also time the fused A'A kernel (extra field, not the metric)")
    ap.add_argument("--lsqr", type=int, default=0, help="also run this many LSQR iterations on b = A x_true (extra field, not the metric)")
    ap.add_argument("--cgnr", type=int, default=0, help="N = 1: also run this many iterations of CG on the normal equations through the fused A'A (extra field)")
    ap.add_argument("--cgls", type=int, default=0, help="N = 1: also run this many CGLS iterations on b = A x_true (extra field, not the metric)")
    ap.add_argument("--placement", choices=("probe", "none"), default="probe",
                    help="probe: the coefficient slab and the range vector come from Jets.stream_pair -- candidate allocations are measured (forward + "
                         "adjoint of the operator itself, outside every timed region) and the best ordered pair is kept; none: plain allocation order")
    ap.add_argument("--fwd-walk", type=int, default=-1, help="pin the tall forward's grid walk to candidate K (0..9; 8, 9: column bands, tried by operators of fewer than 1024 rows) instead of measuring it lazily (profiling one instantiation)")
    ap.add_argument("--placement-candidates", type=int, default=3, help="allocations measured by --placement probe (2: both orders of two slabs)")
    ap.add_argument("--mode", choices=("auto", "ranks", "team"), default=os.environ.get("BENCH_MODE", "auto"),
                    help="N > 1: one worker process per GPU (ranks), ONE worker driving all GPUs (team), or ranks with a team fallback (auto)")
    ap.add_argument("--check", action="store_true", help="team mode: verify replicas and the adjoint after the timed region (extra field)")
    ap.add_argument("--dump", type=str, default="", help="team mode: save member 0's adjoint result to this .npy file")
    return ap.parse_args()


# ---- heartbeats: a worker appends "<unix time> <phase>" lines to its file; the supervisor reads the last one -------------
_HB_PATH = os.environ.get("BENCH_HEARTBEAT")


def beat(phase: str) -> None:
    if _HB_PATH:
        try:
            with open(_HB_PATH, "a") as f:
                f.write(f"{time.time():.3f} {phase}\n")
        except OSError:
            pass


def cpu_baseline(edge: int, nblocks_full: int, sample_blocks: int, pairs: int) -> dict:
    """The CPU oracle's JetBlock_df!/df'! (reference loop structure incl. per-call temporaries,
    oracle/jets_oracle_body.inc) on a bounded sample, single thread."""
    import numpy as np
    from oracle import jets_oracle as jo

    n = edge ** 3
    a = [jo.rng_u01(np.float32, 1, 0, i * n, n) for i in range(sample_blocks)]
    m = jo.rng_u01(np.float32, 2, 0, 0, n)
    d = [jo.rng_u01(np.float32, 3, 0, i * n, n) for i in range(sample_blocks)]
    mt = np.zeros(n, dtype=np.float32)
    ops = [[jo.Block("diag", n, coeff=ai)] for ai in a]
    jo.block_df(ops, d, [m])
    jo.block_df_adj(ops, [mt], d)  # warm-up pair
    times = []
    for _ in range(pairs):
        t0 = time.perf_counter()
        jo.block_df(ops, d, [m])
        jo.block_df_adj(ops, [mt], d)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    bytes_pair = (4 * sample_blocks * n + 2 * n) * 4
    allcores = None
    try:   # separately labelled: NOT the reference's structure (it is single-threaded), same results bit for bit
        a2 = [np.empty(n, dtype=np.float32) for _ in range(sample_blocks)]     # untouched pages: first touched by the threads that use them
        d2 = [np.empty(n, dtype=np.float32) for _ in range(sample_blocks)]
        jo.fill_u01_omp_f32(a2, 1)
        jo.fill_u01_omp_f32(d2, 3)
        assert a2[-1][-7:].tobytes() == a[-1][-7:].tobytes()                 # the same values as the single-thread sample
        mt2 = np.zeros(n, dtype=np.float32)
        t0 = time.perf_counter()
        nt = jo.tall_diag_pair_omp_f32(a2, m, d2, mt2)
        if time.perf_counter() - t0 > 4 * med + 1.0:
            raise RuntimeError("OpenMP run slower than the single-thread run; skipped")
        assert mt2.tobytes() == mt.tobytes(), "all-cores adjoint differs from the single-thread loop"
        tt = []
        for _ in range(max(pairs, 7)):
            t0 = time.perf_counter()
            jo.tall_diag_pair_omp_f32(a2, m, d2, mt2)
            tt.append(time.perf_counter() - t0)
        tt.sort()
        mo = tt[len(tt) // 2]
        scale = sample_blocks / nblocks_full
        allcores = {"value": (1.0 / mo) * scale, "unit": "pairs/s", "cores": nt, "kind": "port, OpenMP over element chunks (NUMA first-touch), rows in order inside a chunk",
                    "spread": {"runs": len(tt), "min": scale / tt[-1], "median": scale / mo, "max": scale / tt[0]},
                    "threads": f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')} (pinned)",
                    "sample": f"same sample, median of {len(tt)} = {mo:.4f} s/pair, {bytes_pair / mo / 1e9:.1f} GB/s algorithmic"}
    except Exception as e:
        allcores = {"value": None, "sample": f"failed: {e!r}"}
    return {
        "value": (1.0 / med) * sample_blocks / nblocks_full,
        "unit": "pairs/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample_blocks} of {nblocks_full} block rows ({edge}^3 Float32 each), median of {pairs} pairs = {med:.3f} s/pair, "
                  f"{bytes_pair / med / 1e9:.1f} GB/s algorithmic; value = sample pairs/s x {sample_blocks}/{nblocks_full} (bandwidth-bound, linear in rows)",
        "build": "gcc -O2 -ftree-vectorize -ffp-contract=off (oracle/Makefile; BASELINE.md section 3)",
        "host_cores_available": os.cpu_count(), "host_cores_granted": CPU_SHARE,
        "all_cores_variant": allcores,
    }


# ---- the supervisor: GPU-free, starts the workers, watches their heartbeats, falls back to team mode ----------------------
FIRST_COLLECTIVE = "first-collective-done"
# phases in which a healthy worker may legitimately be silent for minutes: python / torch being paged in on a fresh box, and the FIRST
# contact with RCCL (topology detection, kernel loading, channel set-up: seconds on most boxes, minutes on some -- one box of this pool
# took 6 minutes over a one-rank RCCL test that takes 11 s elsewhere)
SLOW_PHASES = ("spawned", "python-started", "torch-imported", "process-group-init", "library-imported", "contexts-created", "team-formed",
               "data-resident")
IMPORT_PHASES = SLOW_PHASES


def _last_beat(path):
    """(unix time, phase) of the last complete heartbeat line, or None."""
    try:
        with open(path) as f:
            lines = [ln for ln in f.read().split("\n") if ln.strip()]
        t, _, phase = lines[-1].partition(" ")
        return float(t), phase
    except (OSError, IndexError, ValueError):
        return None


def _phases(path):
    try:
        with open(path) as f:
            return [ln.partition(" ")[2] for ln in f.read().split("\n") if ln.strip()]
    except OSError:
        return []


class Supervisor:
    """One per started process.  `my_ranks` are the workers this process starts (all N for a bare launch, its own rank under
    torch.distributed.run); the LEADER (the process that owns rank 0) reads every rank's heartbeat file, decides, and -- in auto
    mode -- starts the team-mode child.  Supervisors of one job share `hbdir` (same node) and talk through marker files."""

    def __init__(self, args, world, my_ranks, hbdir, leader, base_env, launched_by):
        self.args, self.world, self.my_ranks, self.hbdir, self.leader = args, world, list(my_ranks), hbdir, leader
        self.base_env, self.launched_by = base_env, launched_by
        self.limit = float(os.environ.get("BENCH_WATCHDOG_S", "120"))
        self.limit_import = float(os.environ.get("BENCH_WATCHDOG_IMPORT_S", "300"))
        self.procs = {}
        os.makedirs(hbdir, exist_ok=True)

    def limit_for(self, phase):
        """Seconds without a heartbeat that mean 'stalled' in this phase.  A phase that is ONE long call by construction -- a whole solve
        after beat('lsqr'), the fence behind `--steps` asynchronously enqueued pairs -- gets a limit that grows with the work asked for
        (a generous bound per unit: a pair takes 0.05 s and a solver iteration 0.04 s at the headline size), so that a healthy long run
        is not reported as stalled; a stuck collective still is."""
        if phase in IMPORT_PHASES:
            return self.limit_import
        unit = float(os.environ.get("BENCH_WATCHDOG_UNIT_S", "2"))
        a = self.args
        if phase in ("lsqr", "cgls", "cgnr"):
            return max(self.limit, unit * max(getattr(a, "lsqr", 0) or 0, getattr(a, "cgls", 0) or 0, getattr(a, "cgnr", 0) or 0, 1))
        if phase.startswith("step ") or phase in ("warmup-done", FIRST_COLLECTIVE, "data-resident"):
            return max(self.limit, unit * max(getattr(a, "steps", 0) or 0, 1))
        return self.limit

    # -- files
    def hb(self, r):
        return os.path.join(self.hbdir, f"rank{r}.hb")

    def marker(self, name):
        return os.path.join(self.hbdir, name)

    def write_marker(self, name, text=""):
        tmp = self.marker(name) + ".tmp%d" % os.getpid()
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, self.marker(name))

    def read_marker(self, name):
        try:
            with open(self.marker(name)) as f:
                return f.read()
        except OSError:
            return None

    # -- children: exactly the PIDs started here, never anything by pattern
    def start_worker(self, r, env_extra, whole_env=False):
        env = dict(env_extra) if whole_env else dict(self.base_env, **env_extra)
        env["BENCH_WORKER"] = "1"
        env["BENCH_HEARTBEAT"] = self.hb(r)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL's intra-node transport needs it on this pool
        if not _USER_SET_OMP_THREADS:
            env["OMP_NUM_THREADS"] = "1"                                 # N workers share the host: no CPU work is timed at N > 1
        with open(self.hb(r), "a") as f:
            f.write(f"{time.time():.3f} spawned\n")
        self.procs[r] = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env)

    def stop_children(self, grace=10.0):
        for p in self.procs.values():
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + grace
        for p in self.procs.values():
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                try:
                    p.wait(timeout=5)
                except subprocess.TimeoutExpired:
                    pass

    def say(self, msg):
        print(f"bench.py[supervisor{' 0' if self.leader else ''}]: {msg}", file=sys.stderr, flush=True)

    # -- the ranks phase.  Returns (rc, reason, reached_first_collective)
    def run_ranks(self, rank_env):
        for r in self.my_ranks:
            self.start_worker(r, dict(rank_env(r), BENCH_LAUNCH_NOTE=json.dumps({"launched_by": self.launched_by, "mode_requested": self.args.mode})))
        live = set(self.my_ranks)
        failure = None
        t_own_done = time.time()

        def others_running():                                              # the leader stays until EVERY rank has said 'worker-done' (or one has failed)
            if not self.leader:
                return False
            return any((_last_beat(self.hb(q)) or (0, ""))[1] != "worker-done" for q in range(self.world))

        while (live or others_running()) and failure is None:
            for r in sorted(live):
                code = self.procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    lb = _last_beat(self.hb(r))
                    failure = f"rank {r} exited with {code} in phase '{lb[1] if lb else '?'}'"
                    with open(self.hb(r), "a") as f:
                        f.write(f"{time.time():.3f} DIED rc={code}\n")
            now = time.time()
            if live:
                t_own_done = now
            if failure is None and self.leader:
                for r in range(self.world):                            # the leader sees every rank's file (one node)
                    lb = _last_beat(self.hb(r))
                    if lb is None:
                        if not live and now - t_own_done > self.limit:     # its own ranks are through and rank r has never written a line
                            failure = f"rank {r} never reported (no heartbeat file {self.limit:.0f} s after the leader's own ranks finished)"
                            break
                        continue
                    if lb[1].startswith("DIED"):
                        failure = f"rank {r} died ({lb[1]}; last phase '{([p for p in _phases(self.hb(r)) if not p.startswith('DIED')] or ['?'])[-1]}')"
                    elif lb[1] != "worker-done" and now - lb[0] > self.limit_for(lb[1]):
                        failure = f"rank {r} stalled: no heartbeat for {now - lb[0]:.0f} s in phase '{lb[1]}'"
                    if failure is not None:
                        break
                if failure is not None:                                # where every rank was: a stuck collective shows as everybody waiting for one
                    rows = [(q, _last_beat(self.hb(q))) for q in range(self.world)]
                    failure += "; all ranks: " + ", ".join(f"{q}:'{b[1]}' {now - b[0]:.0f}s ago" if b else f"{q}:no heartbeat" for q, b in rows)
            if failure is None and not self.leader and self.read_marker("abort") is not None:
                failure = "the leader aborted the ranks phase: " + (self.read_marker("abort") or "")
            if failure is None:
                time.sleep(0.05)
        if failure is None:
            return 0, None, True
        reached = any(FIRST_COLLECTIVE in _phases(self.hb(r)) for r in range(self.world))
        if self.leader:
            self.write_marker("abort", failure)
        self.stop_children()
        self.write_marker(f"gone{min(self.my_ranks)}", "")
        return 1, failure, reached

    def run_team_child(self, note):
        env = {k: v for k, v in self.base_env.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                                   "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        env["BENCH_TEAM_CHILD"] = "1"
        env["BENCH_LAUNCH_NOTE"] = json.dumps(note)
        self.procs = {}
        self.start_worker(0, env, whole_env=True)                          # (none of the launcher's rank variables reaches the team worker)
        p = self.procs[0]
        while True:
            code = p.poll()
            if code is not None:
                return code, (None if code == 0 else f"the team-mode child exited with {code} in phase '{(_last_beat(self.hb(0)) or (0, '?'))[1]}'")
            lb = _last_beat(self.hb(0))
            now = time.time()
            if lb and now - lb[0] > self.limit_for(lb[1]):
                self.stop_children()
                return 1, f"the team-mode child stalled: no heartbeat for {now - lb[0]:.0f} s in phase '{lb[1]}'"
            time.sleep(0.05)

    def visible_devices(self):
        """How many devices a worker would see -- asked of a short-lived CHILD process (jh_device_count through ctypes, no torch),
        so this process stays GPU-free.  Only the leader asks: N supervisors probing at once would be N processes on the card."""
        forced = os.environ.get("BENCH_TEST_NDEV")
        if forced is not None:
            return int(forced)
        code = ("import ctypes, sys\nlib = ctypes.CDLL(sys.argv[1])\nn = ctypes.c_int(0)\nrc = lib.jh_device_count(ctypes.byref(n))\n"
                "print(n.value if rc == 0 else 0)\n")
        libpath = os.environ.get("JETSHIP_LIB", os.path.join(ROOT, "jets.jl_amd", "libjetship.so"))
        try:
            out = subprocess.run([sys.executable, "-c", code, libpath], capture_output=True, text=True, timeout=180, env=self.base_env)
            return int(out.stdout.strip().splitlines()[-1])
        except Exception as e:
            self.say(f"could not count the visible devices ({e!r}); assuming one per rank")
            return self.world

    def plan(self, note):
        """ranks / team / refuse, decided BEFORE any worker exists: starting N GPU-holding processes where they cannot work is not
        always recoverable (a box that caps the processes on its cards kills the whole job, supervisor included)."""
        mode, backend = self.args.mode, os.environ.get("BENCH_BACKEND", "nccl")
        if mode == "team":
            return "team"
        ndev = self.visible_devices()
        note["visible_devices"] = ndev
        if ndev <= 0:
            self.say("no MI355X visible -- there is no CPU path to fall back to")
            return "refuse"
        if backend != "nccl" or ndev >= self.world:                    # (gloo: validation runs with several ranks on one device)
            return "ranks"
        why = f"{ndev} device(s) visible for {self.world} ranks: RCCL wants one device per rank, so no rank process was started"
        if mode == "ranks":
            self.say(why + " (--mode ranks: not falling back; --mode team or auto runs the single-process team)")
            return "refuse"
        if ndev != 1:
            self.say(why + "; a single-process team needs one device per member too (or exactly one device for the functional check)")
            return "refuse"
        self.say(why + "; running ONE process with one context per member instead (team mode)")
        note["fallback"] = {"from": "ranks", "reason": why}
        return "team"

    def run(self, rank_env):
        import signal

        signal.signal(signal.SIGTERM, lambda *a: (self.stop_children(2.0), sys.exit(143)))
        mode = self.args.mode
        try:
            if not self.leader:
                return self.follow(rank_env, mode)
            note = {"launched_by": self.launched_by, "mode_requested": mode}
            plan = self.plan(note)
            self.write_marker("plan", plan)
            if plan == "refuse":
                self.write_marker("done", "1")
                return 1
            if plan == "ranks":
                rc, reason, reached = self.run_ranks(rank_env)
                if rc == 0:
                    self.write_marker("done", "0")
                    return 0
                self.say(reason)
                if mode == "ranks" or reached:
                    self.say("no fallback: " + ("--mode ranks" if mode == "ranks" else "the first collective had completed, so this is not a launch problem"))
                    self.write_marker("done", "1")
                    return 1
                deadline = time.time() + 30                              # the other supervisors stop their workers first
                firsts = {0} if len(self.my_ranks) == self.world else set(range(self.world))
                while time.time() < deadline and not all(os.path.exists(self.marker(f"gone{r}")) for r in firsts):
                    time.sleep(0.1)
                self.say("falling back to ONE fresh process driving all GPUs (team mode)")
                for r in range(self.world):
                    try:
                        os.replace(self.hb(r), self.hb(r) + ".ranks")    # kept for the diagnostics, out of the watchdog's way
                    except OSError:
                        pass
                note["fallback"] = {"from": "ranks", "reason": reason}
            rc, reason = self.run_team_child(note)
            if rc != 0:
                self.say(reason)
            self.write_marker("done", str(rc))
            return rc
        except KeyboardInterrupt:
            self.stop_children(2.0)
            return 130

    def follow(self, rank_env, mode):
        """A non-leader under torch.distributed.run: run its own rank, and if the leader aborts the ranks phase (or the mode is
        team from the start) wait for the leader's outcome."""
        deadline = time.time() + 240
        while self.read_marker("plan") is None and time.time() < deadline:
            time.sleep(0.05)
        plan = self.read_marker("plan")
        if plan is None:
            self.say("the leader never published a plan")
            return 1
        if plan == "ranks":
            rc, reason, _ = self.run_ranks(rank_env)
            if rc == 0:
                return 0
            self.say(reason)
            if mode == "ranks":
                return 1
        deadline = time.time() + float(os.environ.get("BENCH_FOLLOW_S", "1500"))
        while time.time() < deadline:
            done = self.read_marker("done")
            if done is not None:
                return 0 if done.strip() == "0" else 1
            time.sleep(0.2)
        self.say("the leader never reported an outcome")
        return 1


def supervise(args) -> int:
    import socket
    import tempfile

    n = args.gpus
    if "RANK" in os.environ:                                           # one of N copies under torch.distributed.run
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ["RANK"])
        if world != n:
            raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {n}")
        hbdir = os.path.join(tempfile.gettempdir(), f"bench_hb_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
        sup = Supervisor(args, world, [rank], hbdir, rank == 0, dict(os.environ), "torch.distributed.run")
        rc = sup.run(lambda r: {})
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        hbdir = tempfile.mkdtemp(prefix="bench_hb_")
        sup = Supervisor(args, n, range(n), hbdir, True, dict(os.environ), "bare python (self-spawned)")
        rc = sup.run(lambda r: dict(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                                    MASTER_PORT=str(port), GROUP_RANK="0", BENCH_SELF_SPAWNED="1"))
    if sup.leader and rc == 0 and os.environ.get("BENCH_KEEP_HB") != "1":
        import shutil

        time.sleep(2.0 if "RANK" in os.environ else 0)                  # the followers read "done" first
        shutil.rmtree(hbdir, ignore_errors=True)
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and os.environ.get("BENCH_WORKER") != "1":
        raise SystemExit(supervise(args))                             # the started process never touches the GPU
    beat("python-started")
    if os.environ.get("BENCH_TEST_FAKE_WORKER") == "1" and os.environ.get("BENCH_WORKER") == "1":
        return fake_worker(args)
    if os.environ.get("BENCH_TEAM_CHILD") == "1" or (args.mode == "team" and "RANK" not in os.environ):
        return worker_team(args)
    return worker_ranks(args)


def fake_worker(args):
    """Test scaffolding for the SUPERVISOR only (tests/test_bench_supervisor.py, CPU): a worker that goes through the phases,
    fails or stalls where the test says, and prints a line that cannot be mistaken for a measurement.  No GPU, no torch."""
    team = os.environ.get("BENCH_TEAM_CHILD") == "1"
    rank = 0 if team else int(os.environ.get("RANK", "0"))
    who = "team" if team else str(rank)
    if os.environ.get("BENCH_TEST_FAIL_BEFORE_COLLECTIVE") == who:
        raise SystemExit(f"worker {who}: BENCH_TEST_FAIL_BEFORE_COLLECTIVE")
    if os.environ.get("BENCH_TEST_STALL_BEFORE_COLLECTIVE") == who:
        beat("process-group-init")
        time.sleep(3600)
    time.sleep(0.3)
    if not team:                                                       # a collective completes only when EVERY rank has entered it
        beat("at-first-collective")
        world, hbdir = int(os.environ.get("WORLD_SIZE", "1")), os.path.dirname(_HB_PATH)
        while not all("at-first-collective" in _phases(os.path.join(hbdir, f"rank{q}.hb")) for q in range(world)):
            time.sleep(0.05)
    beat(FIRST_COLLECTIVE)
    if os.environ.get("BENCH_TEST_FAIL_AFTER_COLLECTIVE") == who:
        raise SystemExit(f"worker {who}: BENCH_TEST_FAIL_AFTER_COLLECTIVE")
    time.sleep(0.3)
    if os.environ.get("BENCH_TEST_LONG_SOLVE_S"):                      # one long call after beat('lsqr'): healthy, silent
        beat("lsqr")
        time.sleep(float(os.environ["BENCH_TEST_LONG_SOLVE_S"]))
    if os.environ.get("BENCH_TEST_SLOW_RANK") == who:                  # a rank that is still working when rank 0 has finished
        time.sleep(float(os.environ.get("BENCH_TEST_SLOW_RANK_S", "3")))
        if os.environ.get("BENCH_TEST_SLOW_RANK_FAILS") == "1":
            raise SystemExit(f"worker {who}: BENCH_TEST_SLOW_RANK_FAILS")
    if rank == 0:
        out = {"metric": "FAKE (supervisor test, nothing was measured)", "value": None, "n_gpus": args.gpus}
        out.update(launch_note("team" if team else "ranks"))
        print(json.dumps(out), flush=True)
    beat("worker-done")


def launch_note(default_mode: str) -> dict:
    """What the supervisor tells its worker about how the job was launched (goes into the JSON line)."""
    try:
        note = json.loads(os.environ.get("BENCH_LAUNCH_NOTE", "") or "{}")
    except ValueError:
        note = {}
    out = {"launch_mode": default_mode}
    if note.get("launched_by"):
        out["launched_by"] = note["launched_by"]
    if note.get("fallback"):
        out["launch_fallback"] = note["fallback"]
    return out


def worker_ranks(args):
    """One process per GPU (torch.distributed, backend nccl = RCCL): this rank's rows of the operator."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch

    beat("torch-imported")
    if os.environ.get("BENCH_TEST_FAIL_BEFORE_COLLECTIVE") == str(rank):   # tests: a rank that dies before the first collective
        raise SystemExit(f"rank {rank}: BENCH_TEST_FAIL_BEFORE_COLLECTIVE")
    if os.environ.get("BENCH_TEST_STALL_BEFORE_COLLECTIVE") == str(rank):  # tests: a rank stuck in "RCCL bootstrap"
        time.sleep(3600)
    force_dist = os.environ.get("BENCH_FORCE_DIST", "0") == "1"      # run the RCCL path even with one rank (validation)
    ndev = torch.cuda.device_count()                                  # (does not initialise the GPU)
    if ndev < 1:
        raise SystemExit("bench.py: no MI355X visible -- there is no CPU path to fall back to")
    backend = os.environ.get("BENCH_BACKEND", "nccl")                # "gloo": validation with several ranks on ONE GPU (RCCL refuses that)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if backend == "nccl" and world > 1 and ndev < local_world and ndev != 1:
        raise SystemExit(f"bench.py: {local_world} ranks on this node but only {ndev} devices visible (RCCL wants one device per rank)")
    if backend == "nccl" and world > 1 and ndev == 1 and os.environ.get("HIP_VISIBLE_DEVICES") is None and os.environ.get("ROCR_VISIBLE_DEVICES") is None:
        raise SystemExit(f"bench.py: --gpus {world} but ONE device visible: RCCL refuses several ranks on one device "
                         "(BENCH_BACKEND=gloo validates the multi-rank flow on a one-GPU box)")
    device = local_rank % ndev                                        # a launcher that narrows HIP_VISIBLE_DEVICES per rank leaves one device
    dist = None
    if world > 1 or (force_dist and "RANK" in os.environ):
        import torch.distributed as dist  # noqa: F811

        torch.cuda.set_device(device)
        beat("process-group-init")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
        probe = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(probe)                                         # RCCL's channels are up on every rank once this returns
        if backend == "nccl":
            torch.cuda.synchronize()
        assert int(probe.item()) == world
        beat(FIRST_COLLECTIVE)

    import jets_jl_amd as J

    J.init(device)
    beat("library-init")
    if args.tune:
        J.tune(**{k: int(v) for k, v in (kv.split("=") for kv in args.tune.split(","))})

    edge, nblocks = args.edge, args.nblocks
    n = edge ** 3
    part = J.rowpart.partition_rows(nblocks, world, rank)
    nloc = part.count

    # ---- operator + vectors, resident in HBM ------------------------------------------------------
    blk = J.JetSpace("float32", edge, edge, edge)
    Rloc = J.JetBSpace([blk] * nloc)
    # Which allocation holds the coefficients and which the range vector is measured, not left to the order of two hipMallocs: on this chip
    # a kernel that reads one 64 GiB slab and writes another runs up to 10 % apart between the two directions and between slabs, differently
    # in every process (jets.jl_amd/placement.py, profiles/exp_r03_swap_roles.txt).  Same values either way (seeded by element index).
    spare = []                                                             # world == 1: the candidates not kept stay for the allocation-order leg below
    if args.placement == "probe":
        # (with a solver section on the command line nothing spare is kept: a third live 64 GiB vector beside a solver's own copy of b
        # can exhaust the device, and the allocation-order leg is then left out -- round-4 advisor finding)
        solver_sections = bool(args.lsqr or args.cgls or args.cgnr)
        coeff, d, placement = J.stream_pair(Rloc, candidates=args.placement_candidates, keep_all=(world == 1 and not solver_sections))
        spare = placement.pop("all", [])
        for k, v in enumerate(spare):                                      # the leg needs candidates 0 and 1 only
            if k >= 2 and v is not coeff and v is not d:
                v.close()
        J.rand_(coeff, seed=1, stream=0, index_base=part.first * n)        # all diagonals of this rank: one slab
        J.rand_(d, seed=3, stream=0, index_base=part.first * n)
    else:
        coeff = J.rand(Rloc, seed=1, stream=0, index_base=part.first * n)
        d = J.rand(Rloc, seed=3, stream=0, index_base=part.first * n)
        placement = {"probed": False}
    A = J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays])
    if args.fwd_walk >= 0:
        J.op_tune_set(A, "fwd_walk", args.fwd_walk)
    m = J.rand(J.domain(A), seed=2, stream=0)
    mt = J.zeros(J.domain(A))
    shard = J.rowpart.for_device(part, A) if dist is not None else None
    J.synchronize()
    beat("data-resident")

    def forward():
        J.mul_(d, A, m)

    def adjoint():
        if shard is not None:
            shard.mul_adj_(mt, d, force_collective=force_dist)
        else:
            J.mul_(mt, A.H, d)

    # operator setup, outside the warm-up count: the first forwards of a large operator each try one grid walk (lazy autotune,
    # jh_tall.hip: lazy_next -- no extra launches, no host sync; 16 calls, 20 with a play-off between the two best, until the choice is made) and the first
    # collective builds RCCL's channels -- with --warmup 0 neither may land in the timed region
    forward()
    adjoint()
    setup_forwards = 1
    while J.op_tune_get(A, "fwd_walk") == -1 and 0 < J.op_tune_get(A, "fwd_trials") and setup_forwards < 32:
        forward()
        J.synchronize()
        setup_forwards += 1
    for _ in range(args.warmup):
        forward()
        adjoint()
    J.synchronize()
    beat("warmup-done")

    ev = [[J.Event() for _ in range(3)] for _ in range(args.steps)]

    def fence():
        J.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        forward()
        ev[k][1].record()
        adjoint()
        ev[k][2].record()
        if k % 5 == 4:
            beat(f"step {k + 1}")                                      # (an append to a local file: microseconds, host side only)
    fence()
    elapsed = time.perf_counter() - t0
    beat("timed-region-done")

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- multi-rank diagnostics, OUTSIDE the timed region: per-rank device times, the local adjoint kernel alone and the
    # stand-alone all-reduce, so that the exposed part of the exchange can be read off the line (adj_ms - adj_kernel_ms)
    multi = None
    if dist is not None:
        reps = max(3, min(args.steps, 10))
        f_ms = sum(ev[k][0].elapsed_ms(ev[k][1]) for k in range(args.steps)) / args.steps
        a_ms = sum(ev[k][1].elapsed_ms(ev[k][2]) for k in range(args.steps)) / args.steps
        J.mul_(mt, A.H, d)
        e0 = J.Event().record()
        for _ in range(reps):
            J.mul_(mt, A.H, d)                                       # this rank's rows, no exchange
        e1 = J.Event().record()
        k_ms = e0.elapsed_ms(e1) / reps
        fence()
        shard.comm.all_reduce_sum_(mt, force=force_dist)
        e0 = J.Event().record()
        for _ in range(reps):
            shard.comm.all_reduce_sum_(mt, force=force_dist)         # the whole 64 MiB domain vector in one piece, nothing to hide behind
        e1 = J.Event().record()
        ar_ms = e0.elapsed_ms(e1) / reps
        on_gpu = dist.get_backend() == "nccl"
        mine = torch.tensor([f_ms, a_ms, k_ms, ar_ms], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rows = [[float(x) for x in t.cpu()] for t in every]
        ar_bytes = n * 4
        ar_worst = max(r[3] for r in rows)
        multi = {
            "backend": dist.get_backend(), "rccl_nranks": world if on_gpu else None,
            "per_rank": [{"rank": r, "rows": J.rowpart.partition_rows(nblocks, world, r).count, "fwd_ms": rows[r][0], "adj_ms": rows[r][1],
                          "adj_kernel_ms": rows[r][2], "exposed_exchange_ms": rows[r][1] - rows[r][2]} for r in range(world)],
            "allreduce": {"bytes": ar_bytes, "chunks_in_adjoint": int(os.environ.get("JETS_AR_CHUNKS", "4")), "ms_standalone": ar_worst,
                          "busbw_GBps": (2.0 * (world - 1) / world) * ar_bytes / ar_worst / 1e6 if world > 1 and ar_worst > 0 else None,
                          "exposed_ms_max": max(r[1] - r[2] for r in rows)},
        }

    fwd_t = [ev[k][0].elapsed_ms(ev[k][1]) for k in range(args.steps)]
    adj_t = [ev[k][1].elapsed_ms(ev[k][2]) for k in range(args.steps)]
    pair_t = sorted(f + a for f, a in zip(fwd_t, adj_t))
    fwd_ms, adj_ms = sum(fwd_t) / args.steps, sum(adj_t) / args.steps
    s = 4
    fwd_bytes = (2 * nloc * n + n) * s        # read a, read m, write d      
    adj_bytes = (2 * nloc * n + n) * s        # read a, read d, write m
    pair_bytes_global = (4 * nblocks * n + 2 * n) * s
    kernels = {
        "forward": {"kernel": "k_tall_diag_fwd", "ms": fwd_ms, "bytes": fwd_bytes, "GBps": fwd_bytes / fwd_ms / 1e6},
        "adjoint": {"kernel": "k_tall_diag_adj" + ("+allreduce" if dist is not None else ""), "ms": adj_ms, "bytes": adj_bytes,
                    "GBps": adj_bytes / adj_ms / 1e6},
    }
    # a tall adjoint beyond 48 GiB is walked in launches of 512 rows (same bits, 2-3 % faster): report per LAUNCH
    adj_launches = max(1, J.tune_get("last_adj_launches")) if dist is None else 1
    kernels["adjoint"]["launches_per_call"] = adj_launches
    kernels["forward"]["launches_per_call"] = 1
    dom = max(kernels.values(), key=lambda kv: kv["ms"])
    traffic, traffic_round = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("nblocks") == nloc and tj.get("edge") == edge:
                key = dom["kernel"].split("+")[0]
                if key == "k_tall_diag_fwd":                            # PMC traffic is recorded per forward-walk CANDIDATE (round 4: one
                    cand = J.op_tune_get(A, "fwd_walk")                 # instantiation per pass, tools/prof_walk_pmc.sh); rounds 1-2 keyed the
                    if cand >= 0 and f"{key}@walk{cand}#round" in tj and tj[f"{key}@walk{cand}#round"] >= "r04":   # grid ORDER (0 sequential / 1 concurrent)
                        key += "@walk%d" % cand
                    else:
                        persistent = J.tune_get("last_fwd_rows_per_wg") >= nloc   # the column-persistent walk re-reads no m, like the row-concurrent one
                        key += "@walk%d" % (1 if persistent else J.tune_get("last_fwd_walk"))
                traffic = tj.get(key)
                if traffic is None:
                    key = key.split("@")[0]                              # counters taken over the walks the lazy autotune tried
                    traffic = tj.get(key)
                traffic_round = tj.get(key + "#round", "r01")
                if traffic is not None and dom["launches_per_call"] > 1 and traffic > 0.75 * dom["bytes"]:
                    traffic /= dom["launches_per_call"]                  # recorded when the whole call was one launch
        except Exception:
            traffic = None

    extra = {}
    if args.fused_normal and world == 1:
        y = J.zeros(J.domain(A))
        C = A.H @ A
        J.mul_(y, C, m)
        e0, e1 = J.Event().record(), None
        for _ in range(args.steps):
            J.mul_(y, C, m)
        e1 = J.Event().record()
        ms = e0.elapsed_ms(e1) / args.steps
        nb = (nloc * n + 2 * n) * s
        extra["fused_normal"] = {"ms": ms, "bytes": nb, "GBps": nb / ms / 1e6, "unfused_pair_ms": fwd_ms + adj_ms}

    if args.lsqr:
        x_true = J.rand(J.domain(A), seed=4, stream=0)
        J.mul_(d, A, x_true)                                   # b = A x_true (this rank's rows), in the range vector's storage
        # N > 1: the whole distributed loop behind the C ABI (jh_lsqr_solve_partitioned over the ABI's own RCCL communicator:
        # ranged steps, ranged all-reduces on the exchange stream, one host synchronisation per step) when every rank can set it
        # up; otherwise the Python driver over torch.distributed.  BENCH_LSQR_ABI=0 forces the latter.
        target, lsqr_driver = (shard if shard is not None else A), ("python driver over torch.distributed" if shard is not None else "jh_lsqr_solve")
        abi_mode = os.environ.get("BENCH_LSQR_ABI", "1")               # "force": also with ONE rank (validation / timing on a one-GPU box)
        if dist is not None and dist.get_backend() == "nccl" and (abi_mode == "force" or (world > 1 and abi_mode == "1")):
            from jets_jl_amd._ffi import lib as _lib

            ok = torch.tensor([1 if _lib.jh_comm_available() == 0 else 0], device="cuda")   # loads librccl, nothing else
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                def exchange_id(raw):
                    box = [raw]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]

                abi = J.rowpart.AbiComm(world, rank, exchange_id=exchange_id)
                if world == 1:
                    J.tune(force_dist=1)                             # run the exchange with the one-rank communicator
                target, lsqr_driver = J.rowpart.for_device(part, A, comm=abi), "jh_lsqr_solve_partitioned (C ABI communicator, pipelined exchange)"
        fence()
        beat("lsqr")
        t_l = time.perf_counter()
        res = J.lsqr(target, d, atol=0.0, btol=0.0, conlim=0.0, maxiter=args.lsqr, overwrite_b=True, force_maxiter=True)
        fence()
        t_l = time.perf_counter() - t_l
        if dist is not None:
            tt = torch.tensor([t_l], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_l = float(tt.item())
        err = (res.x - x_true).materialize()
        one_pass = os.environ.get("JETS_LSQR_FUSED_STEP", "1") != "0"
        # whole job per iteration.  one pass (jh_blockop_bidiag_step): read a, read u, write u = 3Nn, + v, w and the x/w/v updates
        # (~12n per rank); two halves: forward 3Nn+n, adjoint 2Nn+2n, updates ~8n per rank
        it_bytes = ((3 if one_pass else 5) * nblocks * n + (12 if one_pass else 11) * n * world) * s
        extra["lsqr"] = {"iterations": res.itn, "ms_per_iteration": 1e3 * t_l / max(res.itn, 1), "algorithmic_bytes_per_iteration": it_bytes,
                         "GBps": it_bytes * res.itn / t_l / 1e9, "rel_err_vs_x_true": float(J.norm(err)) / float(J.norm(x_true)),
                         "istop": res.istop, "r1norm_first_last": [res.history[0][1], res.history[-1][1]], "driver": lsqr_driver,
                         "schedule": ("one pass per iteration (jh_blockop_bidiag_step): u<-Av-(alpha/beta)u, ||u||^2 and A'u together; v<-A'u/beta-beta v on domain-sized vectors"
                                      if one_pass else "two fused halves: u<-Av-(alpha/beta)u with ||u||^2 (jh_blockop_mul_axpby), v<-A'u/beta-beta v with ||v||^2 (jh_blockop_mul_adj_axpby)")
                                     + "; u never normalised in memory"}

    if args.cgls and world == 1:
        x_true = J.rand(J.domain(A), seed=4, stream=0)
        J.mul_(d, A, x_true)
        fence()
        beat("cgls")
        t_c = time.perf_counter()
        res = J.cgls(A, d, atol=0.0, btol=0.0, maxiter=args.cgls, overwrite_b=True, force_maxiter=True)
        fence()
        t_c = time.perf_counter() - t_c
        err = (res.x - x_true).materialize()
        it_bytes = (4 * nblocks * n + 14 * n) * s                   # pass 1: the coefficients (fused A'A) ; pass 2: a, r in, r out ; ~14 domain-sized streams
        extra["cgls"] = {"iterations": res.itn, "ms_per_iteration": 1e3 * t_c / max(res.itn, 1), "algorithmic_bytes_per_iteration": it_bytes,
                         "GBps": it_bytes * res.itn / t_c / 1e9, "rel_err_vs_x_true": float(J.norm(err)) / float(J.norm(x_true)), "istop": res.istop,
                         "driver": "jh_cgls_solve", "schedule": "two passes per iteration, no range-sized temporary: ||A p||^2 = <p, A'A p> (jh_blockop_normal_mul, N n s bytes), "
                                                                "then r <- r - alpha A p, ||r||^2 and A'r in one pass of the step kernel (3 N n s)"}

    if args.cgnr and world == 1:
        x_true = J.rand(J.domain(A), seed=4, stream=0)
        J.mul_(d, A, x_true)
        fence()
        beat("cgnr")
        t_c = time.perf_counter()
        res = J.cgnr(A, d, atol=0.0, btol=0.0, maxiter=args.cgnr, force_maxiter=True)
        fence()
        t_c = time.perf_counter() - t_c
        err = (res.x - x_true).materialize()
        it_bytes = (nblocks * n + 12 * n) * s                       # the coefficients once ; ~12 domain-sized streams
        extra["cgnr"] = {"iterations": res.itn, "ms_per_iteration": 1e3 * t_c / max(res.itn, 1), "ms_total_incl_the_adjoint_pass_for_A'b": 1e3 * t_c,
                         "algorithmic_bytes_per_iteration": it_bytes, "GBps": it_bytes * res.itn / t_c / 1e9,
                         "rel_err_vs_x_true": float(J.norm(err)) / float(J.norm(x_true)), "istop": res.istop, "driver": "jh_cgnr_solve",
                         "schedule": "CG on (A'A) x = A'b: one adjoint pass for A'b, then ONE fused A'A pass per iteration (jh_blockop_normal_mul: N n s bytes); b only read"}

    # ---- world == 1, after everything that needs the data: the SAME pair in plain allocation order ---------------------------------
    # The headline's operands sit where the probe put them; a caller who allocates coefficients and range vector one after the other
    # and never asks (zeros(range(A)), A*m) gets the process's first two allocations in that order.  One extra run -- the first candidate
    # as the coefficient slab, the second as the range vector, its own operator (and lazy walk measurement), warm-up, then the same
    # number of timed steps -- outside the headline's timed region, so that both figures are on the driver's line.
    placement_none = None
    # what the HEADLINE's forward ran, read before the leg below launches anything (the context's last_* knobs describe the latest launch)
    fwd_grid_walk_hl = ("column-persistent (a workgroup streams every block row)" if J.tune_get("last_fwd_rows_per_wg") >= nloc
                        else {0: "sequential row sweep", 1: "all rows concurrent"}.get(J.tune_get("last_fwd_walk"), "banded"))
    if world == 1 and placement.get("probed") and len(spare) >= 2:
        fwd_walk_hl = J.op_tune_get(A, "fwd_walk")
        if placement.get("kept") == [0, 1]:
            placement_none = {"value": args.steps / elapsed, "unit": "pairs/s", "same_run_as_headline": True,
                              "note": "the probe kept the allocation order: the headline IS the allocation-order figure"}
        else:
            J.close(A)
            c0, d0 = spare[0], spare[1]
            J.rand_(c0, seed=1, stream=0, index_base=part.first * n)
            J.rand_(d0, seed=3, stream=0, index_base=part.first * n)
            A0 = J.blockop([[J.JopDiagonal(c)] for c in c0.arrays])
            if args.fwd_walk >= 0:
                J.op_tune_set(A0, "fwd_walk", args.fwd_walk)
            J.mul_(d0, A0, m)
            J.mul_(mt, A0.H, d0)
            k0 = 1
            while J.op_tune_get(A0, "fwd_walk") == -1 and 0 < J.op_tune_get(A0, "fwd_trials") and k0 < 32:
                J.mul_(d0, A0, m)
                J.synchronize()
                k0 += 1
            for _ in range(args.warmup):
                J.mul_(d0, A0, m)
                J.mul_(mt, A0.H, d0)
            fence()
            t0n = time.perf_counter()
            for _ in range(args.steps):
                J.mul_(d0, A0, m)
                J.mul_(mt, A0.H, d0)
            fence()
            t_none = time.perf_counter() - t0n
            placement_none = {"value": args.steps / t_none, "unit": "pairs/s", "ms_per_step": 1e3 * t_none / args.steps, "steps": args.steps,
                              "same_run_as_headline": False, "fwd_walk_candidate": J.op_tune_get(A0, "fwd_walk"),
                              "note": "coefficients in the process's first 64 GiB allocation, range vector in its second (what --placement none "
                                      "does from the start); timed after the headline, same process, same warm-up and step count"}
            A, A0 = A0, None                                          # (the line below reports the headline's walk: kept above)
        placement_none["headline_fwd_walk_candidate"] = fwd_walk_hl
    else:
        fwd_walk_hl = J.op_tune_get(A, "fwd_walk")
    if rank == 0:
        pairs_per_s = args.steps / elapsed
        out = {
            "metric": "fwd+adj mul! pairs/sec, 1024-block tall JopBlock (256^3 Float32 diagonal blocks)",
            "value": pairs_per_s,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_device": {"median": pair_t[len(pair_t) // 2], "min": pair_t[0], "max": pair_t[-1]},   # HIP events, this rank
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic U[0,1) Float32 from the counter-based generator (seeds a=1, m=2, d=3), generated on device",
            "config": {
                "workload": f"{nblocks}x1 tall JopBlock of diagonal JopLn, {edge}^3 Float32 blocks, fwd+adj mul! pair",
                "nblocks": nblocks, "block": [edge, edge, edge], "rows_per_gpu": nloc,
                "parallelism": f"row-partition x{world}" + (f" + {os.environ.get('BENCH_BACKEND', 'RCCL')} all-reduce({n * s / 2**20:.0f} MiB, pipelined in 4 chunks) in adjoint" if world > 1 else ""),
                "fwd_grid_walk": fwd_grid_walk_hl,
                "fwd_walk_choice": {"candidate": fwd_walk_hl, "setup_forward_calls": setup_forwards},
                "placement": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in placement.items()},
                "tune": {k: J.tune_get(k) for k in ("fwd_group", "fwd_unroll", "fwd_wg", "fwd_order", "adj_unroll", "adj_depth", "adj_wg", "nt", "autotune")},
            },
            "achieved_GBps_pair": pair_bytes_global * pairs_per_s / 1e9,
            "roofline_frac_pair": pair_bytes_global * pairs_per_s / 1e9 / (HBM_PEAK_GBS * world),
            "roofline": {
                "bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": (None if traffic is None else
                                   {"measured_in_this_run": False, "file": "profiles/traffic_latest.json", "round": traffic_round,
                                    "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this same command (tools/prof_default_pmc.sh), "
                                           "(2*FETCH_SIZE + WRITE_SIZE) * 1024 B per launch of the named kernel (gfx950 corrections of MI355X_MICROARCH.md)"}),
                "bytes_per_launch": dom["bytes"] / dom["launches_per_call"], "ms_per_launch": dom["ms"] / dom["launches_per_call"],
                "launches_per_call": dom["launches_per_call"],
            },
            "kernels": kernels,
        }
        out.update(extra)
        if placement_none is not None:
            out["placement_none"] = placement_none
        if world > 1:
            out.update(launch_note("ranks: one process per GPU, torch.distributed over " + ("RCCL" if backend == "nccl" else backend)))
        if multi is not None:
            out["multi_gpu"] = multi
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(edge, nblocks, min(args.cpu_sample_blocks, nblocks), args.cpu_pairs)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 1, "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    beat("worker-done")


def worker_team(args):
    """ONE process drives all N GPUs: one context per device, jh_comm_init_all (ncclCommInitAll), every member's kernels enqueued
    by this host thread, the ranged all-reduces of a range as one group (rowpart.Team / TeamOp).  Same workload, same timed
    region, same JSON line as the one-process-per-GPU form.  With ONE visible device the members are N streams of that device
    (the grouped sum is then a device kernel): the whole flow minus RCCL -- a functional check, not a scaling measurement."""
    import numpy as np

    import jets_jl_amd as J
    from jets_jl_amd import rowpart
    from jets_jl_amd._ffi import check, lib

    beat("library-imported")
    N, edge, nblocks = args.gpus, args.edge, args.nblocks
    n = edge ** 3
    ndev = J.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no MI355X visible -- there is no CPU path to fall back to")
    if ndev >= N:
        ctxs = []
        for dev in range(N):
            J.init(dev)
            ctxs.append(J.context_current()[0])
        one_device = False
        placement = f"{N} device(s), one context each, " + ("RCCL over xGMI (ncclCommInitAll)" if N > 1 else "RCCL team of one (ncclCommInitAll)")
    elif ndev == 1:
        J.init(0)
        ctxs = [J.context_current()[0]] + [J.context_create(0) for _ in range(N - 1)]
        one_device = True
        placement = f"{N} contexts (streams) of ONE device, device-side sum instead of RCCL: functional check of the flow, NOT a scaling measurement"
    else:
        raise SystemExit(f"bench.py: --gpus {N} in team mode needs {N} visible devices (or exactly one, for the functional check); {ndev} are visible")
    if args.tune:
        for c in ctxs:
            with J.using_context(c):
                J.tune(**{k: int(v) for k, v in (kv.split("=") for kv in args.tune.split(","))})
    beat("contexts-created")
    team = rowpart.Team(ctxs)
    beat("team-formed")

    blk = J.JetSpace("float32", edge, edge, edge)
    parts = [rowpart.partition_rows(nblocks, N, k) for k in range(N)]
    ops, coeffs, m_list, d_list, placements = [], [], [], [], []
    for k, _ in team.each():
        Rk = J.JetBSpace([blk] * parts[k].count)
        if args.placement == "probe":                                    # (as in worker_ranks: the better ordered pair of candidate allocations)
            coeff, dk, pl = J.stream_pair(Rk, candidates=args.placement_candidates)
            J.rand_(coeff, seed=1, stream=0, index_base=parts[k].first * n)
            J.rand_(dk, seed=3, stream=0, index_base=parts[k].first * n)
        else:
            coeff = J.rand(Rk, seed=1, stream=0, index_base=parts[k].first * n)
            dk = J.rand(Rk, seed=3, stream=0, index_base=parts[k].first * n)
            pl = {"probed": False}
        placements.append({kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in pl.items() if kk != "pair_ms_other"})
        coeffs.append(coeff)
        ops.append(J.blockop([[J.JopDiagonal(c)] for c in coeff.arrays]))
        m_list.append(J.rand(blk, seed=2, stream=0))
        d_list.append(dk)
    T = team.operator(ops)
    m, d = rowpart.TeamVec(m_list), rowpart.TeamVec(d_list)
    mt = team.zeros(T.domain())
    team.synchronize()
    beat("data-resident")

    def forward():
        T.mul_(d, m)

    def adjoint():
        T.mul_adj_(mt, d)

    def group_sum(vec):                                                  # the whole domain vector in one piece, on the library streams
        with team.group():
            for k in range(N):
                check(lib.jh_comm_allreduce_sum(vec[k].handle))

    forward()
    adjoint()
    team.synchronize()
    beat(FIRST_COLLECTIVE)
    setup_forwards = 1
    while any(J.op_tune_get(A, "fwd_walk") == -1 and 0 < J.op_tune_get(A, "fwd_trials") for A in ops) and setup_forwards < 32:
        forward()
        team.synchronize()
        setup_forwards += 1
    for _ in range(args.warmup):
        forward()
        adjoint()
    team.synchronize()
    beat("warmup-done")

    def events():
        out = []
        for _ in team.each():
            out.append(J.Event())
        return out

    def record(evs):
        for e in evs:
            e.record()                                                   # (an event records on ITS context's stream)

    ev = [[events() for _ in range(3)] for _ in range(args.steps)]
    team.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        record(ev[k][0])
        forward()
        record(ev[k][1])
        adjoint()
        record(ev[k][2])
        if k % 5 == 4:
            beat(f"step {k + 1}")
    team.synchronize()
    elapsed = time.perf_counter() - t0
    beat("timed-region-done")

    # outside the timed region: what ONE pair costs this host thread to enqueue (no synchronisation inside the bracket).  All N
    # members' launches come from here, so this must stay well below the pair's device time (projected 5.7 ms at 8 GPUs) --
    # with the member loop behind one ABI call per operator application (jh_team_mul / jh_team_mul_adj), and, for comparison,
    # spelled out call by call from Python as rounds 2-3 did
    def enqueue_ms():
        samples = []
        for _ in range(7):
            team.synchronize()
            t_e = time.perf_counter()
            forward()
            adjoint()
            samples.append(1e3 * (time.perf_counter() - t_e))
        team.synchronize()
        return sorted(samples)[len(samples) // 2]

    host_enqueue = {"host_enqueue_ms_per_pair": enqueue_ms(), "one_abi_call_per_application": bool(T.one_call)}
    if T.one_call:
        T.one_call = False
        host_enqueue["host_enqueue_ms_per_pair_call_by_call"] = enqueue_ms()
        T.one_call = True
    s = 4
    per = []
    reps = max(3, min(args.steps, 10))
    tmp = team.zeros(T.domain())
    for k, _ in team.each():                                             # outside the timed region: the local adjoint kernel alone, member by member
        J.mul_(tmp[k], ops[k].H, d[k])
        e0 = J.Event().record()
        for _ in range(reps):
            J.mul_(tmp[k], ops[k].H, d[k])
        e1 = J.Event().record()
        per.append({"rank": k, "rows": parts[k].count,
                    "fwd_ms": sum(ev[i][0][k].elapsed_ms(ev[i][1][k]) for i in range(args.steps)) / args.steps,
                    "adj_ms": sum(ev[i][1][k].elapsed_ms(ev[i][2][k]) for i in range(args.steps)) / args.steps,
                    "adj_kernel_ms": e0.elapsed_ms(e1) / reps})
        per[-1]["exposed_exchange_ms"] = per[-1]["adj_ms"] - per[-1]["adj_kernel_ms"]
    team.synchronize()
    scratch = team.zeros(T.domain())
    group_sum(scratch)
    team.synchronize()
    e0 = events()
    record(e0)
    for _ in range(reps):
        group_sum(scratch)
    e1 = events()
    record(e1)
    ar_ms = max(e0[k].elapsed_ms(e1[k]) for k in range(N)) / reps
    ar_bytes = n * s
    multi = {"backend": "device-side sum kernel (one device)" if one_device else "rccl (ncclCommInitAll, single process)",
             "rccl_nranks": None if one_device else N, "placement": placement, "per_rank": per, **host_enqueue,
             "allreduce": {"bytes": ar_bytes, "chunks_in_adjoint": T.nchunks, "ms_standalone": ar_ms,
                           "busbw_GBps": (2.0 * (N - 1) / N) * ar_bytes / ar_ms / 1e6 if N > 1 and ar_ms > 0 and not one_device else None,
                           "exposed_ms_max": max(r["exposed_exchange_ms"] for r in per)}}

    extra = {}
    if args.check or args.dump:
        # adjoint = sum over ALL rows (src/Jets.jl:1045-1053 summed over the members): every replica the same bits, and within
        # 1e-5 (rel l2) of the fp64 sum of the members' ordered partial sums (tmp holds them: the local kernels above)
        forward()
        adjoint()
        team.synchronize()
        got = [mt[k].to_numpy().ravel(order="F") for k in range(N)]
        truth = np.zeros(n, dtype=np.float64)
        for k in range(N):
            truth += tmp[k].to_numpy().ravel(order="F").astype(np.float64)
        rel = float(np.linalg.norm(got[0].astype(np.float64) - truth) / np.linalg.norm(truth))
        extra["check"] = {"replicas_bit_identical": all(g.tobytes() == got[0].tobytes() for g in got[1:]), "adjoint_rel_l2_vs_fp64_sum_of_partials": rel,
                          "tolerance": 1e-5, "ok": bool(rel <= 1e-5 and all(g.tobytes() == got[0].tobytes() for g in got[1:]))}
        if args.dump:
            np.save(args.dump, got[0])

    if args.lsqr:
        x_true = rowpart.TeamVec([J.rand(blk, seed=4, stream=0) for _ in team.each()])
        T.mul_(d, x_true)                                                # b = A x_true, member by member, in the range vectors' storage
        team.synchronize()
        beat("lsqr")
        t_l = time.perf_counter()
        res = J.lsqr(T, d, atol=0.0, btol=0.0, conlim=0.0, maxiter=args.lsqr, overwrite_b=True, force_maxiter=True)
        team.synchronize()
        t_l = time.perf_counter() - t_l
        err = (res.x[0] - x_true[0]).materialize()
        it_bytes = (3 * nblocks * n + 12 * n * N) * s
        extra["lsqr"] = {"iterations": res.itn, "ms_per_iteration": 1e3 * t_l / max(res.itn, 1), "algorithmic_bytes_per_iteration": it_bytes,
                         "GBps": it_bytes * res.itn / t_l / 1e9, "rel_err_vs_x_true": float(J.norm(err)) / float(J.norm(x_true[0])),
                         "istop": res.istop, "driver": "jh_lsqr_solve_team (one call, all members)"}

    pairs_per_s = args.steps / elapsed
    pair_bytes_global = (4 * nblocks * n + 2 * n) * s
    fwd_ms = max(r["fwd_ms"] for r in per)
    adj_ms = max(r["adj_ms"] for r in per)
    nloc = parts[0].count
    fwd_bytes = adj_bytes = (2 * nloc * n + n) * s
    kernels = {"forward": {"kernel": "k_tall_diag_fwd", "ms": fwd_ms, "bytes": fwd_bytes, "GBps": fwd_bytes / fwd_ms / 1e6, "launches_per_call": 1},
               "adjoint": {"kernel": "k_tall_diag_adj+allreduce", "ms": adj_ms, "bytes": adj_bytes, "GBps": adj_bytes / adj_ms / 1e6, "launches_per_call": T.nchunks}}
    dom = max(kernels.values(), key=lambda kv: kv["ms"])
    pair0 = sorted(ev[i][0][0].elapsed_ms(ev[i][2][0]) for i in range(args.steps))
    out = {
        "metric": "fwd+adj mul! pairs/sec, 1024-block tall JopBlock (256^3 Float32 diagonal blocks)",
        "value": pairs_per_s, "unit": "pairs/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_device": {"median": pair0[len(pair0) // 2], "min": pair0[0], "max": pair0[-1]},
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic U[0,1) Float32 from the counter-based generator (seeds a=1, m=2, d=3), generated on device",
        "config": {"workload": f"{nblocks}x1 tall JopBlock of diagonal JopLn, {edge}^3 Float32 blocks, fwd+adj mul! pair",
                   "nblocks": nblocks, "block": [edge, edge, edge], "rows_per_gpu": nloc,
                   "parallelism": f"row-partition x{N}, single-process team: {placement}; all-reduce({n * s / 2**20:.0f} MiB, pipelined in {T.nchunks} ranges) in adjoint",
                   "placement": placements,
                   "fwd_walk_choice": {"candidates": [J.op_tune_get(A, "fwd_walk") for A in ops], "setup_forward_calls": setup_forwards}},
        "achieved_GBps_pair": pair_bytes_global * pairs_per_s / 1e9,
        "roofline_frac_pair": pair_bytes_global * pairs_per_s / 1e9 / (HBM_PEAK_GBS * (1 if one_device else N)),
        "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS,
                     "traffic": None, "bytes_per_launch": dom["bytes"] / dom["launches_per_call"], "ms_per_launch": dom["ms"] / dom["launches_per_call"],
                     "launches_per_call": dom["launches_per_call"], "note": "slowest member; per-member times in multi_gpu.per_rank"},
        "kernels": kernels,
    }
    if one_device and N > 1:
        out["valid_scaling_point"] = False                                # N members time-slicing one GPU
    out.update(extra)
    out.update(launch_note("team: one process, one context per GPU (jh_comm_init_all)"))
    out["launch_mode"] = "team: one process, one context per GPU (jh_comm_init_all)"
    out["multi_gpu"] = multi
    print(json.dumps(out), flush=True)
    beat("line-printed")
    for v in (tmp, scratch):
        v.close()
    team.close()
    beat("worker-done")


if __name__ == "__main__":
    main()
