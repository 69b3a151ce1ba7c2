#!/usr/bin/env python3
"""bench.py -- BiCGSTAB iterations/s + SpMV effective HBM GB/s on the 10M-row CSR workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3], the config the metric is quoted on; fits one GPU):
synthetic random CSR, 1e7 x 1e7, 50 nnz/row, fp64 values / int32 indices, generated in HBM
(SURVEY 8d), b = A x*, x0 = 1, no preconditioner, row-sharded over N GPUs (strong scaling:
the matrix is fixed, each rank owns n/N rows).  A "step" is one BiCGSTAB iteration = both
half steps = 2 SpMV + 3 fused vector kernels (+ 2 all-gathers, 3 all-reduces when N > 1).
The stopping tests are evaluated every step but not taken (CUDAMAT_FLAG_NO_EXIT), and the
solve restarts from x0 = 1 every 50 steps so the residual stays far from underflow; each
restart's extra SpMV is inside the timed region and not counted as a step.

Output: ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_START = time.time()
# thread placement of the CPU baseline, fixed before any OpenMP runtime is loaded; the CPUs this process
# may use are counted first (libgomp binds the initial thread to ONE place once OMP_PROC_BIND is set, after which
# sched_getaffinity reports a single CPU)
try:
    HOST_CPU_SET = set(os.sched_getaffinity(0))
    HOST_CPUS = len(HOST_CPU_SET)
except AttributeError:
    HOST_CPU_SET = None
    HOST_CPUS = os.cpu_count() or 1


def _several_ranks():
    """True for every process of an N > 1 run (launcher, supervisors, workers): none of them runs the CPU baseline"""
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
        return True
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv) and sys.argv[i + 1].isdigit():
            return int(sys.argv[i + 1]) > 1
        if a.startswith("--gpus=") and a[7:].isdigit():
            return int(a[7:]) > 1
    return False


# Only the process that times the CPU baseline (rank 0 at N = 1) asks for bound OpenMP threads: with the variable set,
# libgomp pins the initial thread of EVERY process that loads it to one CPU, and the children a launcher or supervisor
# starts inherit that one-CPU mask (a two-rank rehearsal showed `cpus_of_process: 1` in each worker).
if not _several_ranks():
    os.environ.setdefault("OMP_PROC_BIND", "spread")


def unpin_main_thread():
    """OMP_PROC_BIND makes libgomp bind the INITIAL thread to one CPU the moment the library is loaded -- which `import torch`
    does -- and every thread created afterwards inherits that mask: the HIP runtime's helpers and the library's uploader
    thread would all share CPU 0 with the main thread.  On a box whose CPU 0 is busy that showed as uploads at 11-18 GB/s
    instead of 54 and allocation-heavy set-up stages 3-8 x longer (round 5: two of seven boxes; scripts/setup_probe.py, which
    never sets the variable, was fast on every box).  The OpenMP team of the CPU baseline keeps its places; only this thread
    -- and what it starts -- goes back to the CPUs the process was given."""
    if HOST_CPU_SET:
        try:
            os.sched_setaffinity(0, HOST_CPU_SET)
        except OSError:
            pass


def _cpulist(text):
    cpus = set()
    for tok in text.strip().split(","):
        if not tok:
            continue
        a, _, b = tok.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def bind_near_gpu(torch, local_rank):
    """A two-socket host reaches a GPU through ONE socket: host threads (and the pages they first touch: the arrays a drop-in
    call uploads) on the other socket cost the upload two thirds of its rate and make allocation-heavy set-up stages
    several times slower (round 5: uploads at 11-18 GB/s instead of 54 on the boxes whose GPU hangs off the other socket
    than the one the scheduler happened to pick).  This thread -- and what it starts -- moves to the CPUs the kernel lists
    as local to the device, within the set the process was given.  Returns a note for the JSON line."""
    if not HOST_CPU_SET or os.environ.get("CUDAMAT_BENCH_BIND", "near") == "off":
        return None
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        local = _cpulist(open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read()) & HOST_CPU_SET
        node = open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip()
        if not local:
            return None
        os.sched_setaffinity(0, local)
        return {"gpu_pci": bdf, "numa_node": int(node), "cpus_near_gpu": len(local), "cpus_of_process": len(HOST_CPU_SET)}
    except (OSError, ValueError, AttributeError):
        return None



class GpuStateSampler:
    """The device's own core clock, package power and temperature while the timed region runs (sysfs hwmon of its PCI device,
    readable by an ordinary user; 20 samples a second from a thread that only reads three small files).  Why it is in the
    JSON line: the streaming ceilings measured in one process agree to 2 % from box to box, the blocked SpMV's launch time
    does not (2.50-2.73 ms) -- the core clock under the board's power management is the one thing left that differs."""

    def __init__(self, bdf):
        import glob
        import threading
        self.rows, self.stop_flag, self.thread = [], threading.Event(), None
        hw = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf) if bdf else []
        self.files = None
        if hw:
            f = {"sclk_hz": "freq1_input", "power_uw": "power1_input", "temp_mc": "temp2_input", "mem_temp_mc": "temp3_input"}
            self.files = {k: os.path.join(hw[0], v) for k, v in f.items() if os.path.exists(os.path.join(hw[0], v))}
        if self.files:
            self.thread = threading.Thread(target=self._loop, daemon=True)

    def _loop(self):
        while not self.stop_flag.is_set():
            row = {}
            for k, path in self.files.items():
                try:
                    row[k] = float(open(path).read())
                except (OSError, ValueError):
                    pass
            self.rows.append(row)
            time.sleep(0.05)

    def start(self):
        if self.thread:
            self.thread.start()
        return self

    def stop(self):
        if not self.thread:
            return None
        self.stop_flag.set()
        self.thread.join()

        def stat(key, scale):
            v = sorted(r[key] * scale for r in self.rows if key in r)
            return None if not v else {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)}
        return {"samples": len(self.rows), "sclk_mhz": stat("sclk_hz", 1e-6), "power_w": stat("power_uw", 1e-6),
                "junction_c": stat("temp_mc", 1e-3), "hbm_c": stat("mem_temp_mc", 1e-3), "source": "hwmon of the device, sampled during the timed region"}


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max), or None when unlimited / unknown: a one-GPU box of
    the pool shows 256 CPUs and is throttled to 16"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
CHUNK = 50              # steps per restart: the synthetic systems converge by ~0.025 per iteration, so ||r||^2 reaches 1e-160
                        # after 50 steps and would underflow after ~95 -- restart from x0 well before that
CHUNK_SMALL = 100       # ... of the latency-bound small Poisson-type systems (C2 needs ~190 iterations to 1e-8)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="rand50", choices=["rand50", "poisson5", "mat10000"],
                    help="mat10000 = BASELINE configs[1]: the 100x100 5-point Laplacian of mat10000.mtx "
                         "(the generator reproduces the file bit for bit, tests/test_oracle_golden.py)")
    ap.add_argument("--nx", type=int, default=4000, help="grid width of the poisson5 workload")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--per-row", type=int, default=50)
    ap.add_argument("--precond", default="none", choices=["none", "ilu0", "bjilu0"],
                    help="ilu0: the reference's preconditioner (N > 1: independent replicas); bjilu0: block-Jacobi "
                         "ILU(0) of each rank's diagonal block, row-sharded (different maths for N > 1, SURVEY 8 f4)")
    ap.add_argument("--loop", default="pbicgstab", choices=["pbicgstab", "pipelined"],
                    help="pbicgstab: the reference's loop (pbicgstab.cu:45-154), the headline; pipelined: the same "
                         "recurrences re-arranged so that the reductions run beside the SpMVs (SURVEY 8 f4; with --precond: its preconditioned form)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--drop-in", default="auto", choices=["auto", "off"],
                    help="auto: also time the host-pointer entry point (cudamat_solve = bicgstab(), pbicgstab.h:113) end to end, "
                         "twice (one GPU only; the second call reuses the first one's plan)")
    ap.add_argument("--cpu-iters-full", type=int, default=2, help="iterations of the CPU baseline on the full matrix, per OpenMP team size")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--gpu-warm-seconds", type=float, default=3.0,
                    help="untimed iterations of the same loop BEFORE the W warm-up steps, for this long: the first process on an idle "
                         "box finds the GPU in its low-power state and its first seconds of work run 6-8 %% slower (round 5: "
                         "the first bench.py on a fresh box 175-176 it/s, every later one 181-190); 0 = none")
    ap.add_argument("--other-configs", default="auto", choices=["auto", "off"],
                    help="auto: after the judged region of the default (headline) invocation on one GPU, also run BASELINE.json's "
                         "other single-GPU configurations -- C3 poisson5, C5 rand50 + ILU(0), C2 mat10000 -- as side sections "
                         "of the same JSON line (`other_configs`)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# N > 1: who starts the ranks, and what happens when an exchange form hangs
#
#   python bench.py --gpus N              (no WORLD_SIZE)  -> launcher(): starts `python -m torch.distributed.run
#                                                             --nproc-per-node N bench.py ...` as a CHILD and relays rank 0's line
#   torch.distributed.run ... bench.py    (WORLD_SIZE = N) -> supervisor(): every rank process starts its WORKER as a child,
#                                                             one set of fresh workers per exchange form, each with a time limit
#   worker (CUDAMAT_BENCH_WORKER=1)                        -> run_bench(): the only processes that touch a GPU
#
# The inter-GPU exchange forms (grouped ncclSend/ncclRecv pieces behind phase 1, plain RCCL all-gather, torch.distributed)
# cannot be rehearsed on a one-GPU development box, and a form that HANGS cannot be survived inside the process that
# runs it.  So neither the launcher nor the supervisors ever make a GPU call (no torch.cuda.*, no cuda_mat_amd.lib()):
# they only start children (fork + exec before anything touched the GPU), watch them, and kill the process group of a
# set that times out or dies; the next form then gets fresh processes.  The supervisors agree on each set's fate over a
# CPU (gloo) group.  `comm.launcher` in the JSON line records every form that was tried.
# ---------------------------------------------------------------------------------------------------------------
DEFAULT_FORMS = ["rccl:1", "rccl:0", "torch:0"]      # most capable first; the last is slow but needs nothing new


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


FORM_LIMITS = (180.0, 120.0, 120.0)   # s per exchange form, first to last (a healthy two-rank set takes 4-9 s after the first import)
BUDGET_S = 500.0                      # all forms together, from the start of bench.py: the driver ends a bench run after 600 s (which
                                      # include torch.distributed.run's own start-up), and a line must be out before that
FORM_RESERVE_S = 45.0                 # what every LATER form is left with at least when an earlier one runs into its limit


def _form_limit(i, n_forms=None, left=None):
    """seconds a set of workers may take for exchange form i (import, generation, tuning, gate, timed region): the form's
    own limit (CUDAMAT_BENCH_FORM_TIMEOUT pins one for all), shrunk so that the forms after it keep FORM_RESERVE_S each
    inside what is left of the overall budget (`left` seconds; None = no budget)"""
    v = os.environ.get("CUDAMAT_BENCH_FORM_TIMEOUT")
    limit = float(v) if v else FORM_LIMITS[min(i, len(FORM_LIMITS) - 1)]
    if left is not None:
        reserve = float(os.environ.get("CUDAMAT_BENCH_RESERVE") or FORM_RESERVE_S)
        limit = min(limit, left - reserve * max((n_forms or 1) - 1 - i, 0))
    return limit


def _budget():
    v = os.environ.get("CUDAMAT_BENCH_BUDGET")
    return float(v) if v else BUDGET_S


def _kill_group(proc):
    """end the worker (and anything it started): it runs in its own session, so its pid is its process group"""
    import signal
    if proc.poll() is None:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    try:
        proc.wait(timeout=30)
    except Exception:  # noqa: BLE001 - a process stuck in the kernel: go on without it
        pass


def _die_with_parent():
    """(child side, between fork and exec) have the kernel end this process when the supervisor that started it goes --
    a worker lives in its own session, so nothing else would reach it when a supervisor is killed from outside"""
    import ctypes
    import signal
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGKILL))      # PR_SET_PDEATHSIG
    except OSError:
        pass


def _worker_cmd():
    import shlex
    stub = os.environ.get("CUDAMAT_BENCH_WORKER_CMD")           # tests/test_bench_launch_cpu.py: a stand-in worker, no GPU
    return shlex.split(stub) if stub else [sys.executable, os.path.abspath(__file__)]


def supervisor(argv):
    """One rank process of `torch.distributed.run` with WORLD_SIZE > 1.  Makes no GPU call."""
    import subprocess
    import tempfile
    import torch
    import torch.distributed as dist
    unpin_main_thread()      # (an OMP_PROC_BIND from the caller's environment: the workers started below inherit this thread's mask)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    backend = os.environ.get("CUDAMAT_BENCH_BACKEND", "nccl")
    comm_kind = os.environ.get("CUDAMAT_BENCH_COMM", "rccl" if backend == "nccl" else "torch")
    forms = list(DEFAULT_FORMS) if comm_kind == "rccl" else ["torch:1", "torch:0"]
    if os.environ.get("CUDAMAT_BENCH_FORMS"):
        forms = os.environ["CUDAMAT_BENCH_FORMS"].split(",")
    log, line, rc_final = [], None, 1
    # the overall budget runs from the launcher's start (CUDAMAT_BENCH_DEADLINE, epoch seconds) or, under a bare
    # torch.distributed.run, from rank 0's start of this function; every rank uses rank 0's deadline
    box = [float(os.environ.get("CUDAMAT_BENCH_DEADLINE") or (T_START + _budget())) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    deadline = box[0]
    for i, form in enumerate(forms):
        box = [(_free_port(), _form_limit(i, len(forms), deadline - time.time())) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        port, limit = box[0]
        if limit < 5.0:
            log.append({"form": form, "ok": False, "seconds": 0.0, "limit_s": round(limit, 1), "ranks": ["not started: budget spent"] * world})
            if rank == 0:
                print("bench.py supervisor: form %s not started, %.0f s of the budget left" % (form, deadline - time.time()), file=sys.stderr, flush=True)
            continue
        env = dict(os.environ, CUDAMAT_BENCH_WORKER="1", CUDAMAT_BENCH_FORMS=form, MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TORCHELASTIC_USE_AGENT_STORE="False")
        out = tempfile.TemporaryFile(mode="w+")
        t0 = time.time()
        proc = subprocess.Popen(_worker_cmd() + argv, env=env, stdout=out, start_new_session=True, preexec_fn=_die_with_parent)
        while True:
            rc = proc.poll()
            status = 0 if rc is None else (1 if rc == 0 else 2)
            if rc is None and time.time() - t0 > limit:
                status = 3
            t = torch.tensor([float(status), -float(status)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst, best = int(t[0].item()), int(-t[1].item())
            if worst >= 2 or best == 1:
                break
            time.sleep(0.5)
        ok = worst == 1
        if not ok:
            _kill_group(proc)
        outcomes = [None] * world
        dist.all_gather_object(outcomes, {0: "running", 1: "ok", 2: "exit %s" % proc.returncode, 3: "time limit"}[status])
        log.append({"form": form, "ok": ok, "seconds": round(time.time() - t0, 1), "limit_s": round(limit, 1), "ranks": outcomes})
        if rank == 0:
            print("bench.py supervisor: form %s -> %s (%s)" % (form, "ok" if ok else "FAILED", outcomes), file=sys.stderr, flush=True)
        if ok:
            out.seek(0)
            lines = [l for l in out.read().splitlines() if l.startswith("{")]
            line = lines[-1] if lines else None
            rc_final = 0
            break
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        if rc_final == 0 and line is not None:
            res = json.loads(line)
            res.setdefault("comm", {})["launcher"] = log
            # a line timed on a later form of the ladder is NOT the headline configuration: say so at the top level
            res["degraded_form"] = (None if len(log) == 1 else
                                    "timed on exchange form %s after %s failed" % (log[-1]["form"], ", ".join(e["form"] for e in log[:-1])))
            print(json.dumps(res), flush=True)
        else:
            print("bench.py: every exchange form failed: %s" % json.dumps(log), file=sys.stderr, flush=True)
            rc_final = 1
    return rc_final


def launcher(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the ranks as a child job.  Makes no GPU call."""
    import subprocess
    total = _budget() + 40.0           # the supervisors keep to the budget themselves; this is the backstop
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    # this pool's host driver only supports dmabuf IPC: without this switch RCCL (and any sharing of device memory
    # between processes) fails with `hipIpcGetMemHandle: invalid argument`.  The image exports it already; it is set
    # here so that a job started from a scrubbed environment still has it ("launch environment")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("CUDAMAT_BENCH_DEADLINE", "%.1f" % (T_START + _budget()))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=total)
    except subprocess.TimeoutExpired:
        _kill_group(proc)
        print("bench.py: the %d-rank job did not finish within %.0f s" % (args.gpus, total), file=sys.stderr)
        return 1
    lines = [l for l in (out or "").splitlines() if l.startswith("{")]
    if proc.returncode != 0 or not lines:
        print("bench.py: the %d-rank job ended with code %s and %d result lines" % (args.gpus, proc.returncode, len(lines)),
              file=sys.stderr)
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def cpu_teams():
    """OpenMP team sizes the CPU baseline tries: around the CPUs' worth of time this process is GRANTED (cgroup quota; a
    one-GPU box of the pool shows 256 CPUs and is throttled to 16) -- half, once, twice and four times that -- plus 8 /
    16 / 32 / 64, never more than the CPUs the process may run on"""
    q = cpu_quota()
    q = int(round(q)) if q else HOST_CPUS
    cand = {q // 2, q, 2 * q, 4 * q, 8, 16, 32, 64}
    teams = sorted(c for c in cand if 1 <= c <= HOST_CPUS) or [HOST_CPUS]
    return teams[-6:]


def cpu_baseline(args):
    """The reference's CPU path (bicstab_omp BiCG, bicstab.cpp:93-196) as restated by the oracle, faithful threading
    (SpMV + dot OpenMP, the five vector loops serial as upstream), timed on this host ON THE WHOLE WORKLOAD MATRIX
    (built in memory, 6 GB at 1e7 x 50; first touch by the generator's static row partition, which is the
    SpMV's): `value` = iterations / seconds of the reference's iteration loop (bicstab.cpp:146-182; BiCG, like BiCGSTAB,
    costs 2 SpMV per iteration).  The OpenMP team is chosen ON THAT LOOP: the program runs `--cpu-iters-full` iterations
    once per team size of cpu_teams() after one transposition, and `value` is the best of them (ties: fewer threads);
    the whole table is reported (`loop_iters_per_s_by_threads`), so `value` >= every other team's figure by
    construction.  The one-off transposition (Transpose2, a serial loop upstream: ~70 s at this size) is done with
    every thread and reported beside the rate, not in it."""
    from oracle import oracle as O
    unpin_main_thread()          # (the GPU sections ran on the CPUs near the GPU: the baseline gets every CPU the process was given)
    O.set_num_threads(min(HOST_CPUS, 64))
    full_rows = args.rows

    def build(n):
        if args.workload == "poisson5":
            nx = min(args.nx, n)
            return O.poisson5(nx, n // nx)
        return O.rand_rows(n, args.per_row, args.seed)

    t0 = time.perf_counter()
    A = build(full_rows)
    t_build = time.perf_counter() - t0
    n = A.n
    xs = O.xstar(n, args.seed + 1)
    b = O.spmv(A, xs)
    iters_full = max(1, args.cpu_iters_full)
    teams = cpu_teams()
    _, its, t_tr, t_loops = O.bicg_teams(A, b, teams, maxit=iters_full, eps=0.0)
    rate = {t: max(i, 1) / tl for t, i, tl in zip(teams, its, t_loops)}
    t_best = max(teams, key=lambda t: (rate[t], -t))
    # (ii) of SURVEY 8d: the same loop with its five vector loops parallel too (what a tuned host port would do)
    _, its_p, _, t_loops_p = O.bicg_teams(A, b, [t_best], maxit=iters_full, eps=0.0, parallel_vec=True)
    # like for like: the oracle's restatement of the GPU loop (BiCGSTAB, pbicgstab.cu:581-754) on the
    # same matrix; tol = 0 never triggers, so exactly iters_full iterations run
    O.set_num_threads(t_best)
    t1 = time.perf_counter()
    _, _, st2 = O.pbicgstab2(A, b, maxit=iters_full, tol=0.0)
    dt2 = time.perf_counter() - t1
    del A, b, xs
    i_best = teams.index(t_best)
    return {
        "value": rate[t_best], "unit": "iter/s", "cores": t_best, "cores_are": "OpenMP threads used (not physical cores)",
        "cpus_granted": cpu_quota() or HOST_CPUS, "kind": "port",
        "sample": "oracle BiCG restatement of bicstab_omp (2 SpMV/iter; SpMV+dot OpenMP, vector loops serial as in the "
                  "reference): %d iterations of its loop (bicstab.cpp:146-182) on the FULL %d-row x %d nnz/row matrix, "
                  "built in host memory by the same generator; %d threads = the team that runs THIS LOOP fastest (sweep below)"
                  % (its[i_best], n, args.per_row if args.workload == "rand50" else 5, t_best),
        "loop_seconds": t_loops[i_best], "transpose_seconds": t_tr, "build_seconds": t_build,
        "threads": {"nproc": os.cpu_count(), "sched_affinity_at_start": HOST_CPUS, "cgroup_cpu_quota_cores": cpu_quota(),
                    "omp_threads_used": t_best, "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
                    "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES"),
                    "loop_iters_per_s_by_threads": {str(t): rate[t] for t in teams}},
        "parallel_vector_loops": {"value": max(its_p[0], 1) / t_loops_p[0], "unit": "iter/s",
                                  "sample": "the same BiCG loop with its five vector loops under OpenMP as well, %d iterations, %d threads"
                                            % (its_p[0], t_best)},
        "bicgstab_port": {"value": max(st2.iters, 1) / dt2, "unit": "iter/s",
                          "sample": "oracle BiCGSTAB restatement (pbicgstab.cu:581-754), %d iterations on the same full matrix, %d threads"
                                    % (max(st2.iters, 1), t_best)},
    }


def hbm_ceiling(ctx):
    """measured streaming ceilings of this GPU, in-repo kernels on 2^26 doubles per vector (512 MiB each, far
    beyond the 256 MB Infinity Cache), HIP-event timed, median of 7: `gbs` = y += a*x (2 reads + 1 write per element,
    triad-shaped: the ceiling of a read/write mix), `read_gbs` = sum x*y (2 reads: the ceiling of a read-only stream)"""
    from cuda_mat_amd import _lib
    n = 1 << 26
    x, y = ctx.empty(n), ctx.empty(n)
    x.zero()
    y.zero()
    t = ctx.timer()

    def med(fn):
        times = []
        for i in range(9):
            t.start()
            fn()
            t.stop()
            if i >= 2:
                times.append(t.elapsed_ms())
        return sorted(times)[len(times) // 2]

    ms = med(lambda: ctx.axpy(n, 0.5, x, y))
    out = ctx.empty(1)
    ms_r = med(lambda: _lib.check(_lib.lib().cudamat_dot(ctx.h, n, x.ptr, y.ptr, out.ptr)))
    # a write-only and a 1 read : 1 write stream (the runtime's fill and copy kernels), for kernels whose traffic is mostly
    # writes (phase 1 of the blocked SpMV on a value dictionary: 27 % reads, 73 % writes)
    ms_w = med(lambda: x.zero())
    ms_c = med(lambda: _lib.check(_lib.lib().cudamat_d2d(ctx.h, y.ptr, x.ptr, 8 * n)))
    t.close()
    out.free()
    x.free()
    y.free()
    return {"kernel": "k_axpy, 2 reads + 1 write, 3 x 512 MiB", "gbs": 24.0 * n / ms / 1e6,
            "read_kernel": "k_dot, 2 reads, 2 x 512 MiB", "read_gbs": 16.0 * n / ms_r / 1e6,
            "write_kernel": "hipMemsetAsync, 512 MiB", "write_gbs": 8.0 * n / ms_w / 1e6,
            "copy_kernel": "hipMemcpyAsync device to device, 512 MiB", "copy_gbs": 16.0 * n / ms_c / 1e6}


def main():
    args = parse()
    argv = sys.argv[1:]
    is_worker = os.environ.get("CUDAMAT_BENCH_WORKER") == "1"
    world_env = os.environ.get("WORLD_SIZE")
    if not is_worker and os.environ.get("CUDAMAT_BENCH_SUPERVISE", "1") != "0":
        if world_env is None and args.gpus > 1:
            sys.exit(launcher(args, argv))
        if world_env is not None and int(world_env) > 1:
            sys.exit(supervisor(argv))
    if os.environ.get("CUDAMAT_BENCH_ONE_DEVICE"):
        # rehearsal with several ranks on ONE GPU: two dependency-driven (spin-waiting) kernels of different
        # processes may starve each other of workgroup slots, so use the level-by-level triangular solves
        os.environ.setdefault("CUDAMAT_TRSV_SYNCFREE", "0")
    if os.environ.get("CUDAMAT_BENCH_WATCHDOG"):      # dump every thread's Python stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CUDAMAT_BENCH_WATCHDOG"]), exit=True)
    # stdout carries exactly ONE line (the JSON): libraries that chat on fd 1 (RCCL prints its library
    # path when a communicator is created) are sent to stderr until the result is ready
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        out = run_bench(args)
        if os.environ.get("CUDAMAT_BENCH_OTHER_CONFIGS") == "off":      # (scripts that profile or A/B the headline alone)
            args.other_configs = "off"
        headline = (args.workload == "rand50" and args.rows == 10_000_000 and args.per_row == 50 and args.precond == "none"
                    and args.loop == "pbicgstab")
        if out is not None and out.get("n_gpus") == 1 and headline and args.other_configs != "off":
            out["other_configs"] = other_configs(args)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if out is not None:
        print(json.dumps(out), flush=True)


# keys of a full line that a side section keeps (the rest is either the headline's business or repeats the section's name)
SECTION_KEYS = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "spmv_gbs", "spmv_form", "value_dictionary",
                "trsv_ms_per_apply", "trsv_roofline", "trsv_traffic", "levels", "setup_s", "drop_in", "gpu_state", "memory_placement")
OTHER_CONFIGS = (      # BASELINE.json configs[2], [4], [1]: name, argument overrides
    ("poisson5", {"workload": "poisson5", "precond": "none", "steps": 200, "warmup": 10}),
    ("rand50_ilu0", {"workload": "rand50", "precond": "ilu0", "steps": 50, "warmup": 5}),
    ("mat10000", {"workload": "mat10000", "precond": "none", "steps": 2000, "warmup": 200}),
)


def other_configs(args):
    """BASELINE.json's other single-GPU configurations, each run like its own `python bench.py --workload ... --precond ...`
    (generated in HBM, gated by a real solve, warmed up, timed between synchronisations, SpMV launches timed by HIP events)
    AFTER the judged region and outside it: the reference prints its phase times for whatever it is given
    (example.cpp:364-365, pbicgstab.cu:349,362); the driver's one bench run times one configuration, these are the others."""
    import copy
    sections = {}
    for name, over in OTHER_CONFIGS:
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        a.cpu_baseline, a.other_configs, a.loop = "off", "off", "pbicgstab"
        t0 = time.perf_counter()
        try:
            full = run_bench(a)
            sec = {k: full[k] for k in SECTION_KEYS if k in full}
        except (Exception, SystemExit) as e:  # noqa: BLE001 - a side section must never take the headline down
            sec = {"error": "%s: %s" % (type(e).__name__, e)}
        sec["section_seconds"] = time.perf_counter() - t0
        sections[name] = sec
    return sections


def run_bench(args):
    import torch
    unpin_main_thread()
    import cuda_mat_amd as cm
    from cuda_mat_amd.dist import RcclComm, TorchComm, shard_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback exists)"
    # rehearsal knobs (not used by the driver): all ranks on device 0 and/or the gloo backend, so that a
    # one-GPU box can run the real multi-process path (RCCL itself refuses two ranks on one device)
    if os.environ.get("CUDAMAT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("CUDAMAT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    host_placement = bind_near_gpu(torch, local_rank)
    # CUDAMAT_FORCE_SHARDED=1 under torchrun --nproc-per-node 1 exercises the whole N > 1 code path
    # (process group, TorchComm callbacks, sharded loop) on a single GPU
    use_dist = world > 1 or (os.environ.get("CUDAMAT_FORCE_SHARDED") == "1" and "MASTER_ADDR" in os.environ)
    # Collectives of the data path.  "rccl" (default): the library's own RCCL binding, called from the C++ loop
    # (csrc/comm_rccl.hip); torch.distributed then only carries the communicator id, the barriers and the final
    # max over ranks, over a CPU (gloo) group.  "torch": round 1's callbacks into torch.distributed (NCCL = RCCL
    # backend, or gloo for one-GPU rehearsals).
    comm_kind = os.environ.get("CUDAMAT_BENCH_COMM", "rccl" if backend == "nccl" else "torch")
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl" and comm_kind == "torch":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo" if backend == "nccl" else backend)

    def host_allreduce(value, op="sum"):
        """a scalar over the ranks through a CPU tensor (works on every backend)"""
        if not use_dist:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return float(t.item())

    if args.workload == "mat10000":
        args.workload, args.rows, args.nx = "poisson5", 10000, 100
    n = args.rows
    if args.workload == "poisson5":
        nx = args.nx
        ny = n // nx
        n = nx * ny
    # The ILU(0) path does not shard (dependencies cross row blocks; block-Jacobi would be different maths,
    # DESIGN.md section 7): with --precond ilu0 on N > 1 GPUs every rank solves the WHOLE system = N replicas.
    replicas = world > 1 and args.precond == "ilu0"
    row0, row1, per = (0, n, n) if replicas else shard_rows(n, world, rank)
    nloc = row1 - row0

    # The judged numbers are measured on fp64 VALUES (what a matrix of arbitrary coefficients has): the library's value
    # dictionary (8-bit indices when a matrix holds <= 256 distinct values, as SURVEY 8d's generator happens to produce)
    # is switched off for the timed region and reported as a side figure (`with_value_dictionary`).
    forced_fp64 = (("CUDAMAT_VALUE_DICT" not in os.environ or os.environ.get("CUDAMAT_BENCH_FORCED_FP64") == "1")
                   and os.environ.get("CUDAMAT_BENCH_HEADLINE", "fp64") != "dict")
    if forced_fp64:
        os.environ["CUDAMAT_VALUE_DICT"] = "0"
        os.environ["CUDAMAT_BENCH_FORCED_FP64"] = "1"        # (a later section of the same process: the bench set it, not the user)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = cm.Context(local_rank, stream=stream.cuda_stream)
        # the resident solver's one-off placement of its arrays by memory class may take what it takes: on boxes whose
        # allocator runs at 8 GB/s the search's 32 GB of slabs cost 4 s, the library's default budget (0.3 s) would give up, and the
        # loop timed below would run at the luck of the draw (175 ... 190 it/s).  Set-up, reported in `memory_placement.seconds`.
        if "CUDAMAT_PB_PLACE_MAX_MS" not in os.environ:
            ctx.set_option("PB_PLACE_MAX_MS", "20000")
        # ---- generate this rank's row block in HBM
        def make_solver():
            if args.workload == "rand50":
                rn = cm.lib().cudamat_rand_row_nnz(n, args.per_row)
                nnz_ = nloc * rn
                rp = torch.empty(nloc + 1, dtype=torch.int32, device=dev)
                ci = torch.empty(nnz_, dtype=torch.int32, device=dev)
                va = torch.empty(nnz_, dtype=torch.float64, device=dev)
                ctx.gen_rand_rows(n, args.per_row, args.seed, row0, row1, 0, rp, ci, va)
            else:
                rp = torch.empty(nloc + 1, dtype=torch.int32, device=dev)
                tmp_nnz = 5 * nloc
                ci = torch.empty(tmp_nnz, dtype=torch.int32, device=dev)
                va = torch.empty(tmp_nnz, dtype=torch.float64, device=dev)
                ctx.gen_poisson5(nx, ny, row0, row1, 0, rp, ci, va)
                nnz_ = int(rp[-1].item())
            sv = cm.Solver(ctx, nloc, n, nnz_, rp, ci, va, 0)
            del rp, ci, va
            torch.cuda.empty_cache()
            return sv, nnz_

        solver, nnz = make_solver()
        xs = torch.empty(nloc, dtype=torch.float64, device=dev)
        b = torch.empty(nloc, dtype=torch.float64, device=dev)
        x = torch.empty(nloc, dtype=torch.float64, device=dev)
        ctx.gen_xstar(row0, row1, args.seed + 1, xs)
        precond = {"none": cm.PRECOND_NONE, "ilu0": cm.PRECOND_ILU0, "bjilu0": cm.PRECOND_BLOCK_ILU0}[args.precond]
        flags = cm.FLAG_NO_EXIT | cm.FLAG_X0_ONES
        loop = cm.LOOP_PIPELINED if args.loop == "pipelined" else cm.LOOP_PBICGSTAB

        def gate():
            """correctness gate (real stopping rule): b = A x*, solve; the recursive residual the loop reports must be
            the true residual ||b - A x||, and where the solve converges x must equal x*.  Returns (stats, error)."""
            solver.spmv(xs, b)                    # b = A x*
            if precond == cm.PRECOND_ILU0:
                solver.ilu0()
            elif precond:
                solver.block_ilu0()
            st = solver.solve(b, x, precond=precond, loop=loop, maxit=200, tol=1e-8, flags=cm.FLAG_X0_ONES)
            ax = torch.empty(nloc, dtype=torch.float64, device=dev)
            solver.spmv(x, ax)
            res2 = float(((b - ax) ** 2).sum().item())
            if use_dist and not replicas:
                res2 = host_allreduce(res2)
            true_res = res2 ** 0.5
            if not abs(true_res - st.nrm) <= 1e-6 * st.nrm0 + 1e-3 * st.nrm:
                return st, "true residual %g vs loop residual %g" % (true_res, st.nrm)
            err = float((x - xs).abs().max().item())
            if st.converged and not err < 2e-5:          # 1e-5 relative; x* lies in [1, 2)
                return st, "converged but max|x-x*|=%g" % err
            return st, None

        # The exchange forms, most capable first; a form whose gate fails on ANY rank is dropped by ALL ranks (the
        # multi-GPU forms cannot be rehearsed on the one-GPU development box, so the bench checks before it times).
        comm = None
        comm_desc = None
        ladder = [None]
        if use_dist and not replicas:
            # (after the rccl forms the last resort is torch.distributed over the CPU group this process has: slow,
            # but it keeps a scaling run from ending without a number; the JSON names the form that was timed)
            ladder = [("rccl", True), ("rccl", False), ("torch", False)] if comm_kind == "rccl" else [("torch", True), ("torch", False)]
            if os.environ.get("CUDAMAT_BENCH_FORMS"):        # tests: e.g. "torch:0" pins one form
                ladder = [(f.split(":")[0], f.split(":")[1] == "1") for f in os.environ["CUDAMAT_BENCH_FORMS"].split(",")]
        gate_log = []
        st = None
        for form in ladder:
            failure = None
            try:
                if form is not None:
                    kind, overlap = form
                    ctx.set_option("OVERLAP", "1" if overlap else "0")
                    if comm is None or comm_desc[0] != kind:
                        solver.set_comm(None)
                        if comm is not None and hasattr(comm, "close"):
                            comm.close()
                        comm = RcclComm(ctx, rank, world) if kind == "rccl" else TorchComm(device=dev, pieces=True)
                    solver.set_comm(comm.struct)
                    comm_desc = form
                    if os.environ.get("CUDAMAT_BENCH_TEST_HANG") == "%s:%d" % (kind, 1 if overlap else 0):
                        time.sleep(1e6)          # tests: this form "hangs" (the supervisor must replace the workers)
                st, failure = gate()
            except Exception as e:  # noqa: BLE001 - a failing form must not take the bench down
                failure = "%s: %s" % (type(e).__name__, e)
                if os.environ.get("CUDAMAT_BENCH_WORKER") == "1" and use_dist:
                    # a worker under a supervisor: the peers of this rank may be inside a collective that will never
                    # complete -- leave at once (no further collective, no destructors); the supervisors notice the exit
                    # code within a second, end the other workers and give the next form fresh processes
                    print("bench.py worker rank %d: form %s failed: %s" % (rank, form, failure), file=sys.stderr, flush=True)
                    os._exit(3)
            bad = host_allreduce(1.0 if failure else 0.0) if use_dist else (1.0 if failure else 0.0)
            gate_log.append({"form": None if form is None else {"comm": form[0], "overlap": form[1]},
                             "failed_ranks": int(bad), "rank0_failure": failure})
            if bad == 0:
                break
        else:
            raise SystemExit("parity gate failed for every exchange form: %s" % json.dumps(gate_log))
        conv_iters = st.iters if st.converged else None
        # digest of the gate's solution over all ranks: two runs whose exchange forms differ must print the same one
        import hashlib
        digest = hashlib.sha256(x.cpu().numpy().tobytes()).hexdigest()
        if use_dist:
            box = [None] * world
            dist.all_gather_object(box, digest)
            digest = hashlib.sha256("".join(box).encode()).hexdigest()

        timed_steps = [0]      # steps of the timed region that carried per-kernel / per-collective events
        exch = {}              # exchange timings of those steps (cudamat_stats, HIP events inside the C++ loop)

        def run(steps, fl):
            """`steps` iterations in chunks of CHUNK.  With N > 1 ranks only the FIRST chunk carries the HIP events
            around every SpMV and collective (14 event records per iteration are ~3 % of a 1.5 ms sharded iteration);
            on one GPU every chunk does (4 records next to two 2.7 ms SpMVs)."""
            ms_spmv, n_spmv, ms_trsv, n_trsv = 0.0, 0, 0.0, 0
            left = steps
            first = True
            while left > 0:
                # (preconditioned runs converge by ~1e-4 per iteration: 25 steps per restart there)
                k = min(left, CHUNK_SMALL if n <= 200_000 and args.workload != "rand50" else (25 if precond else CHUNK))
                events = first or world == 1
                f = fl if events else fl & ~cm.FLAG_PROFILE
                st = solver.solve(b, x, precond=precond, loop=loop, maxit=k, tol=1e-8, flags=f)
                assert st.iters == k, (st.iters, k)
                ms_spmv += st.ms_spmv
                n_spmv += st.n_spmv
                ms_trsv += st.ms_trsv
                n_trsv += st.n_trsv
                if events and (f & cm.FLAG_PROFILE):
                    timed_steps[0] += k
                    for key in ("ms_gather", "ms_gather_exposed", "ms_allreduce", "n_gather", "n_allreduce"):
                        exch[key] = exch.get(key, 0) + getattr(st, key)
                    exch["overlapped"] = st.overlapped
                left -= k
                first = False
            return ms_spmv, n_spmv, ms_trsv, n_trsv, st

        def barrier():
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        # the GPU out of its idle power state first (see --gpu-warm-seconds), then the contract's W untimed steps
        t_warm0, warm_steps = time.perf_counter(), 0
        while args.gpu_warm_seconds > 0 and time.perf_counter() - t_warm0 < args.gpu_warm_seconds:
            run(25 if precond else CHUNK, flags)
            torch.cuda.synchronize()
            warm_steps += 25 if precond else CHUNK
        if use_dist:
            dist.barrier()
        run(args.warmup, flags)
        timed_steps[0] = 0
        exch.clear()
        barrier()
        sampler = GpuStateSampler((host_placement or {}).get("gpu_pci") if rank == 0 else None).start()
        t0 = time.perf_counter()
        # per-launch SpMV timing = HIP events around every SpMV inside the loop; on L2-resident systems
        # (C2) the four event records per iteration would dominate the ~28 us iteration, and no roofline
        # is quoted for that latency-bound case anyway
        latency_bound = n <= 200_000 or os.environ.get("CUDAMAT_BENCH_NO_EVENTS") == "1"      # (the switch: an A/B of the event records themselves, scripts/events_ab.sh)
        ms_spmv, n_spmv, ms_trsv, n_trsv, st = run(args.steps, flags | (0 if latency_bound else cm.FLAG_PROFILE))
        barrier()
        dt = time.perf_counter() - t0
        gpu_state = sampler.stop()
        dt = host_allreduce(dt, "max")

        # what the output needs to know about the resident solver; it is closed before the drop-in calls below, so that those
        # run like a host program's would -- not beside a second ~50 GB solver of the same matrix -- and draw on the blocks it
        # hands back to the library's pool (csrc/pool.cpp)
        sv_mode, sv_kernel, sv_dict = solver.spmv_mode(), solver.spmv_kernel(), solver.value_dict()
        sv_place = solver.placement()
        sv_place["budget_ms"] = int(os.environ.get("CUDAMAT_PB_PLACE_MAX_MS", "20000"))

        # The value dictionary as a side figure: the same workload timed -- outside the judged region, one GPU only -- with
        # the 8-bit value indices the library would pick by itself for this matrix (<= 256 distinct values)
        side = None
        if (world == 1 and not latency_bound and precond == cm.PRECOND_NONE and forced_fp64
                and os.environ.get("CUDAMAT_BENCH_COMPARE", "1") != "0"):
            ctx.set_option("VALUE_DICT", "1")         # (a context reads the environment once; later changes go through set_option)
            s2 = None
            try:
                s2, _ = make_solver()
                s2.solve(b, x, precond=precond, loop=loop, maxit=3, tol=1e-8, flags=flags)       # selects the form, warms up
                if s2.value_dict() > 0:
                    k2 = min(10, CHUNK)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    st2 = s2.solve(b, x, precond=precond, loop=loop, maxit=k2, tol=1e-8, flags=flags | cm.FLAG_PROFILE)
                    torch.cuda.synchronize()
                    dt2 = time.perf_counter() - t2
                    side = {"value": k2 / dt2, "unit": "iter/s", "steps": k2, "avg_launch_ms": st2.ms_spmv / max(st2.n_spmv, 1),
                            "value_dictionary": s2.value_dict(), "spmv_mode": s2.spmv_mode()}
            except Exception as e:  # noqa: BLE001 - the comparison must never take the bench line down
                side = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                ctx.set_option("VALUE_DICT", "0")
                if s2 is not None:
                    s2.close()

        # The drop-in entry point end to end (the reference's "total delta time" next to its "algorithm delta time",
        # example.cpp:364-365): ONE call of cudamat_solve() on HOST arrays -- upload over PCIe, analysis, loop, download --
        # and a second call with the same matrix, which reuses the first one's plan.  Outside the judged region.
        drop_in = None
        ceil = hbm_ceiling(ctx) if (world == 1 and not latency_bound) else None
        if world == 1 and args.drop_in != "off" and precond in (cm.PRECOND_NONE, cm.PRECOND_ILU0) and not latency_bound:
            solver.set_comm(None) if comm is not None else None
            solver.close()
            solver = None
            try:
                import numpy as np
                from cuda_mat_amd import api as cm_api
                rn = cm.lib().cudamat_rand_row_nnz(n, args.per_row) if args.workload == "rand50" else 5
                rp_d = torch.empty(n + 1, dtype=torch.int32, device=dev)
                ci_d = torch.empty(n * rn, dtype=torch.int32, device=dev)
                va_d = torch.empty(n * rn, dtype=torch.float64, device=dev)
                if args.workload == "rand50":
                    ctx.gen_rand_rows(n, args.per_row, args.seed, 0, n, 0, rp_d, ci_d, va_d)
                else:
                    ctx.gen_poisson5(nx, ny, 0, n, 0, rp_d, ci_d, va_d)
                torch.cuda.synchronize()
                rp_h = rp_d.cpu().numpy()
                nz = int(rp_h[-1])
                ci_h, va_h, b_h = ci_d[:nz].cpu().numpy(), va_d[:nz].cpu().numpy(), b.cpu().numpy()
                del rp_d, ci_d, va_d
                torch.cuda.empty_cache()
                cm.lib().cudamat_plan_cache_clear()
                # one throw-away call on a small system first: what a process pays ONCE (the runtime's pageable-copy path, the
                # call's own context and stream, the uploader thread) is not what a call with a new matrix costs; measured
                # apart by scripts/upload_probe.py (0.16 s after such a call, 0.31-0.40 s as the very first GPU
                # work of a process)
                ws = 4096
                w_rp = (np.arange(ws + 1, dtype=np.int32) * 3).astype(np.int32)
                w_ci = np.clip(np.repeat(np.arange(ws, dtype=np.int32), 3) + np.tile(np.array([-1, 0, 1], np.int32), ws), 0, ws - 1)
                w_ci = np.sort(w_ci.reshape(ws, 3), axis=1)
                w_ci[0] = [0, 1, 2]
                w_ci[-1] = [ws - 3, ws - 2, ws - 1]
                w_va = np.tile(np.array([-1.0, 4.0, -1.0]), ws).reshape(ws, 3)
                w_va[0] = [4.0, -1.0, 0.0]
                w_va[-1] = [0.0, -1.0, 4.0]
                cm_api._solve(ws, 3 * ws, w_va.ravel().copy(), w_rp, w_ci.ravel().astype(np.int32), None, None, np.ones(ws), cm.PRECOND_NONE,
                              cm.LOOP_PBICGSTAB, 200, 1e-8, False)
                cm.lib().cudamat_plan_cache_clear()
                calls = []
                for _ in range(2):
                    t1 = time.perf_counter()
                    xh, sth = cm_api._solve(n, nz, va_h, rp_h, ci_h, None, None, b_h, precond, cm.LOOP_PBICGSTAB, 200, 1e-8, False)
                    wall = time.perf_counter() - t1
                    calls.append({"end_to_end_s": wall, "upload_s": sth.t_upload,
                                  # (the host side of a box shows here: 51-54 GB/s on most boxes of the pool, 11-18 GB/s on some,
                                  # where allocation-heavy set-up stages are slow as well -- DESIGN 6a)
                                  "upload_gbs": (12.0 * nz + 4.0 * (n + 1) + 8.0 * n) / sth.t_upload / 1e9 if sth.t_upload > 0 else None,
                                  "setup_s": sth.t_setup, "tune_s": sth.t_tune,
                                  "analysis_s": sth.t_analysis, "factor_s": sth.t_factor,
                                  "loop_s": sth.t_solve, "library_total_s": sth.t_total, "iters": sth.iters,
                                  "converged": bool(sth.converged), "spmv_mode": sth.spmv_mode, "plan_reused": sth.plan_reused,
                                  "max_abs_err": float(np.abs(xh - xs.cpu().numpy()).max())})
                cm.lib().cudamat_plan_cache_clear()
                drop_in = {"call": "cudamat_solve() on host CSR arrays, tol 1e-8 (= %s; the reference's "
                                   "'total delta time' vs 'algorithm delta time', example.cpp:364-365)"
                                   % ("bicgstab_lu_precond(), pbicgstab.h:119: upload, level analysis, ILU(0), loop" if precond else "bicgstab(), pbicgstab.h:113"),
                           "host_bytes_uploaded": 12.0 * nz + 4.0 * (n + 1) + 8.0 * n,
                           "process_state": "after one throw-away cudamat_solve() on a 4096-row system (start-up costs of the "
                                            "process are not part of a call; scripts/upload_probe.py measures them apart)",
                           "first_call": calls[0], "second_call_same_matrix": calls[1]}
                del rp_h, ci_h, va_h, b_h
            except Exception as e:  # noqa: BLE001 - a side measurement must never take the bench line down
                drop_in = {"error": "%s: %s" % (type(e).__name__, e)}

    its = args.steps / dt * (world if replicas else 1)      # replicas: N independent solves in the same time
    spmv_ms = ms_spmv / max(n_spmv, 1)
    # algorithmic bytes of one local SpMV launch: values+colidx, rowptr, x once, y once
    b_spmv = 12.0 * nnz + 4.0 * (nloc + 1) + 8.0 * n + 8.0 * nloc
    achieved = b_spmv / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    vec_bytes = 144.0 * nloc
    # B_iter(none) = 2 B_spmv + 144 n;  B_iter(ilu0) = B_iter(none) + 2 (12 nnz + 8 (n + 1) + 32 n): the two
    # applications of L^-1 U^-1 per iteration (factor entries + row pointers of both factors + the vectors they stream)
    b_precond = 2.0 * (12.0 * nnz + 8.0 * (nloc + 1) + 32.0 * nloc) if precond else 0.0
    b_iter = 2 * b_spmv + vec_bytes + b_precond
    blocked = sv_mode == 1
    kernel = sv_kernel + (" (one SpMV = the pair)" if blocked else "")
    # small systems: the SpMV rides inside a fused kernel (cudamat_stats.loop_form: 1 = the vector updates folded into the two
    # SpMV launches, 2 = the whole loop in ONE launch) -- name the kernel a trace of the timed region shows
    if st.loop_form == 2:
        kernel = "k_resident_loop<%d> (the whole loop in one launch, grid barriers between its phases; SpMV rows as %s)" % (
            256, sv_kernel)
    elif st.loop_form == 1:
        kernel = "k_fspmv (vector updates folded into the two SpMV launches of an iteration; rows as %s)" % sv_kernel
    if n_spmv == 0:
        kernel += " (L2-resident, launch-latency-bound: per-launch timing off, no roofline quoted)"
    # HBM bytes per SpMV launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # FETCH_SIZE doubled per the gfx950 calibration); only for the exact workload they were taken on
    traffic, traffic_src, traffic_parts = None, None, None
    prof_dir = None
    loop_tag = ""
    if world == 1 and args.rows == 10_000_000 and precond in (cm.PRECOND_NONE, cm.PRECOND_ILU0):
        if args.workload == "rand50" and args.per_row == 50:
            prof_dir = "*ilu0" if precond else "*rand50"
            loop_tag = "[SpMV of the loop]" if precond else ""       # (scripts/summarize_profiles.py splits the blocked kernels of a C5 trace)
        elif args.workload == "poisson5" and args.nx == 4000 and not precond:
            prof_dir = "*poisson5"
    if prof_dir:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", prof_dir, "pmc_fetch_write.json")), reverse=True):
            pm = json.load(open(f))
            # (kernel names as the trace prints them, without argument lists; phase 1 has a dictionary form)
            if blocked:
                p1 = "cm::k_pb_phase1_dict" if sv_dict > 0 else "cm::k_pb_phase1"
                want = [p1, "cm::k_pb_phase2<"]
            else:
                want = ["cm::" + sv_kernel]
            got = [next((v for k, v in pm.items() if (k == w + loop_tag or (w.endswith("<") and k.startswith(w) and k.endswith(loop_tag or ">")))
                         and isinstance(v, dict) and "hbm_bytes_per_launch_corrected" in v), None) for w in want]
            if all(g is not None for g in got):
                traffic = sum(g["hbm_bytes_per_launch_corrected"] for g in got)
                traffic_parts = [g["hbm_bytes_per_launch_corrected"] for g in got]
                traffic_src = os.path.relpath(f, ROOT)
                break
    out = None
    if rank == 0:
        out = {
            # BASELINE.json's metric; `value` is its first component (iterations/s), the second one
            # (SpMV effective HBM GB/s) is `spmv_gbs` = roofline.achieved
            "metric": "BiCGSTAB iters/sec + SpMV effective HBM GB/s, 10M-row CSR at 1/2/4/8 GPUs"
                      if (args.workload == "rand50" and args.rows == 10_000_000 and args.per_row == 50)
                      else "BiCGSTAB iters/sec + SpMV effective HBM GB/s (%s, %d rows)" % (args.workload, n),
            "value": its, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if replicas else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s n=%d nnz=%d%s, x0=1, b=A*xstar, %s, %s x%d"
                                   % (args.workload, n, nnz * world if (args.workload == "rand50" and not replicas) else nnz,
                                      "/rank" if (world > 1 and args.workload != "rand50") else "",
                                      {"none": "no preconditioner", "ilu0": "ILU(0)", "bjilu0": "block-Jacobi ILU(0)"}[args.precond],
                                      "independent replicas" if replicas else "row-sharded", world),
                       "rows": n, "nnz_per_rank": nnz,
                       "parallelism": ("replicas x%d (preconditioned path does not shard)" % world) if replicas else "rows/%d" % world,
                       "converges_in_iters": conv_iters, "gate_x_sha256": digest,
                       "loop": "pbicgstab.cu:45-154" if args.loop == "pbicgstab" else "pipelined BiCGStab (not a reference loop)"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": b_spmv, "avg_launch_ms": spmv_ms,
                         "launches_timed": n_spmv,
                         "iteration_bytes": b_iter,
                         "iteration_bytes_formula": "2 B_spmv + 144 n" + (" + 2 (12 nnz + 8 (n + 1) + 32 n)  [SURVEY 8d, B_iter(ilu0)]" if precond else "  [SURVEY 8d, B_iter(none)]"),
                         "iteration_frac": b_iter * (args.steps / dt) / 1e9 / HBM_PEAK_GBS},
            "spmv_gbs": achieved, "spmv_form": {1: "blocked two-phase", 2: "SELL-C-sigma", 3: "row-pattern dictionary"}.get(
                sv_mode, "csr (lanes-per-row / stream tiles)"),
            # 0: the timed kernels read fp64 values (8 B per entry).  The bench switches the library's value dictionary off
            # (CUDAMAT_VALUE_DICT=0) unless CUDAMAT_BENCH_HEADLINE=dict: SURVEY 8d's generator draws from 39 distinct values,
            # which the 8-bit dictionary form would exploit -- that run is the side figure `with_value_dictionary`
            "value_dictionary": sv_dict,
            # where the blocked copy's arrays went: device memory comes in three classes and the product stream is placed in one of
            # its own (csrc/spmv_pb.hip place_copy; unplaced, the same kernels took 2.50 ... 2.77 ms per launch pair by luck of the draw)
            "memory_placement": sv_place,
            # untimed iterations run before the W warm-up steps to bring the GPU out of its idle power state (--gpu-warm-seconds)
            "gpu_warm": {"seconds": args.gpu_warm_seconds, "steps": warm_steps},
            # where the host side of this process runs: the CPUs local to the GPU's PCIe root (bind_near_gpu)
            "host_placement": host_placement,
            # the device's core clock / power / temperature during the timed region (GpuStateSampler)
            "gpu_state": gpu_state,
        }
        if side is not None:
            if side.get("avg_launch_ms", 0) > 0:
                # EFFECTIVE rate: SURVEY 8d's 12 B per entry over the time of a launch that streams 5 B per entry of
                # matrix data -- a property of this matrix's few distinct values, hence a side figure and never `roofline`
                g2 = b_spmv / (side["avg_launch_ms"] * 1e-3) / 1e9
                side["effective_spmv_gbs"] = g2
                side["effective_frac"] = g2 / HBM_PEAK_GBS
                side["note"] = ("the library's default for this matrix: 8-bit indices into a dictionary of its %d distinct fp64 "
                                "values (csrc/valdict.hip), bit-identical results; `value`/`roofline` above are measured with "
                                "CUDAMAT_VALUE_DICT=0, i.e. on the fp64 values a matrix of arbitrary coefficients has"
                                % side["value_dictionary"])
            out["with_value_dictionary"] = side
        if drop_in is not None:
            out["drop_in"] = drop_in
        if comm is not None:
            # rank 0's exchanges inside the timed region (HIP events recorded by the C++ loop, cudamat_stats): the
            # all-gathers of the SpMV inputs, the part of them rank 0's stream actually waited for ("exposed"; the
            # rest ran behind phase 1 of the blocked SpMV), and the all-reduces of the dot products
            ts = max(timed_steps[0], 1)
            out["comm"] = {"form": {"comm": "rccl (csrc/comm_rccl.hip, called from the C++ loop)" if comm_desc[0] == "rccl"
                                    else "torch.distributed callbacks (%s)" % dist.get_backend(),
                                    "gather": {1: "in pieces, overlapped with phase 1", 2: "windows only (halo): %.4f of a whole gather"
                                                                                             % st.gather_fraction}.get(exch.get("overlapped"), "plain all-gather")},
                           "gather_ms_per_step": exch.get("ms_gather", 0.0) / ts,
                           "gather_exposed_ms_per_step": exch.get("ms_gather_exposed", 0.0) / ts,
                           "gather_hidden_ms_per_step": (exch.get("ms_gather", 0.0) - exch.get("ms_gather_exposed", 0.0)) / ts,
                           "allreduce_ms_per_step": exch.get("ms_allreduce", 0.0) / ts,
                           "gathers_per_step": exch.get("n_gather", 0) / ts, "allreduces_per_step": exch.get("n_allreduce", 0) / ts,
                           "steps_with_events": timed_steps[0], "gate": gate_log}
        if precond:
            out["trsv_ms_per_apply"] = ms_trsv / max(n_trsv / 2, 1)
            if ms_trsv > 0:
                # one application = L^-1 then U^-1: SURVEY 8d's 12*nnz + 8(n+1) + 32*n algorithmic bytes (nnz of this rank)
                b_trsv = 12.0 * nnz + 8.0 * (nloc + 1) + 32.0 * nloc
                gbs = b_trsv / (out["trsv_ms_per_apply"] * 1e-3) / 1e9
                out["trsv_roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_application": b_trsv}
            # fabric bytes per application of L^-1 U^-1, from the committed counter passes of this workload (2 * FETCH_SIZE +
            # WRITE_SIZE per launch): the NEAR part (dependency-driven launches, one per group and factor; an UPPER bound: the
            # correction factor 2 holds for wide coalesced loads and these kernels gather 8 bytes per lane) and the FAR part
            # (the blocked SpMV phases of groups 1.., streams: 28 B per far entry + x tiles)
            if world == 1 and args.workload == "rand50" and args.rows == 10_000_000 and args.per_row == 50 and precond == cm.PRECOND_ILU0:
                import glob
                for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*ilu0", "pmc_fetch_write.json")), reverse=True):
                    pm = json.load(open(f))
                    near = next((v for name, v in pm.items() if name.startswith("cm::k_trsv_syncfree") and isinstance(v, dict)
                                 and "hbm_bytes_per_launch_corrected" in v), None)
                    far = [v for name, v in pm.items() if name.startswith("cm::k_pb_phase") and "far part" in name and isinstance(v, dict)
                           and "hbm_bytes_per_launch_corrected" in v]
                    groups = [st.trsv_groups_l, st.trsv_groups_u]       # what the library's plan holds (cudamat_stats)
                    if near is not None and far and min(groups) >= 2 and st.trsv_form == 1:
                        n_near = sum(groups)
                        n_far = sum(g - 1 for g in groups)            # blocked SpMVs (phase 1 + phase 2 each) per application
                        p1 = [v for name, v in pm.items() if name.startswith("cm::k_pb_phase1") and "far part" in name and isinstance(v, dict)]
                        p2 = [v for name, v in pm.items() if name.startswith("cm::k_pb_phase2") and "far part" in name and isinstance(v, dict)]
                        wavg = lambda vs, key: sum(v[key] * v["launches"] for v in vs) / max(sum(v["launches"] for v in vs), 1)
                        far_bytes = n_far * (wavg(p1, "hbm_bytes_per_launch_corrected") + wavg(p2, "hbm_bytes_per_launch_corrected"))
                        near_bytes = n_near * near["hbm_bytes_per_launch_corrected"]
                        out["trsv_traffic"] = {
                            "near": {"kernel": "k_trsv_syncfree (entries inside a group of levels)", "launches_per_application": n_near,
                                     "bytes_per_launch_upper_bound": near["hbm_bytes_per_launch_corrected"],
                                     "bytes_per_application_upper_bound": near_bytes, "avg_launch_us": near["avg_us"]},
                            "far": {"kernel": "k_pb_phase1 + k_pb_phase2 on the entries whose column lies in an earlier group",
                                    "spmvs_per_application": n_far, "bytes_per_application": far_bytes,
                                    "avg_launch_us": {"phase1": wavg(p1, "avg_us"), "phase2": wavg(p2, "avg_us")}},
                            "bytes_per_application": near_bytes + far_bytes,
                            "vs_algorithmic": (near_bytes + far_bytes) / b_trsv if ms_trsv > 0 else None,
                            "source": os.path.relpath(f, ROOT)}
                        break
            out["levels"] = [st.n_levels_l, st.n_levels_u]
            # one-off setup (outside the timed region): level analysis of L and U; ILU(0) + factor layout + far/near split
            out["setup_s"] = {"analysis": st.t_analysis, "factor": st.t_factor}
        if traffic is not None and spmv_ms > 0:
            # what the fabric really carried per launch (PMC), as a rate: the figure to hold against a stream ceiling
            out["roofline"]["moved_gbs"] = traffic / (spmv_ms * 1e-3) / 1e9
            out["roofline"]["moved_over_algorithmic"] = traffic / b_spmv
        if world == 1 and not latency_bound:
            out["roofline"]["measured_stream_ceiling"] = ceil
            out["roofline"]["frac_of_measured_ceiling"] = achieved / ceil["gbs"]
            if sv_mode == 3:
                # SURVEY 8d's B_spmv prices 4 B of column index per entry; the row-pattern form fetches none (a row's columns come
                # from a 16 KB table through one byte per row), so it MOVES fewer bytes than B_spmv and `achieved` (algorithmic
                # bytes / time) can exceed a stream ceiling measured in moved bytes: compare `moved_gbs` with the ceiling instead
                out["roofline"]["note_algorithmic_vs_moved"] = (
                    "row-pattern SpMV: no column index is fetched, so moved bytes (%s) < algorithmic bytes (%.3g); "
                    "frac_of_measured_ceiling > 1 is the formula's 4 B/entry of indices that never travel -- "
                    "moved_gbs / ceiling = %s" % ("%.3g per launch, PMC" % traffic if traffic else "see profiles/*poisson5",
                                                  b_spmv, "%.2f" % (traffic / (spmv_ms * 1e-3) / 1e9 / ceil["read_gbs"]) if traffic and spmv_ms > 0 else "n/a"))
            if blocked and traffic_parts is not None:
                # what this SpMV's own traffic would take at the two measured ceilings: phase 1 is a read/write mix
                # (10 B read : 8 B written per entry), phase 2 reads only
                t_floor = (traffic_parts[0] / ceil["gbs"] + traffic_parts[1] / ceil["read_gbs"]) / 1e6
                out["roofline"]["ms_at_measured_ceilings"] = t_floor
                out["roofline"]["note_ceilings"] = ("counter bytes of phase 1 / triad ceiling + counter bytes of phase 2 / read ceiling: "
                                                    "the launch pair runs at %.2f of that" % (t_floor / spmv_ms if spmv_ms > 0 else 0.0))
        if world == 1 and args.cpu_baseline != "off":
            out["cpu_baseline"] = cpu_baseline(args)
    if solver is not None:
        solver.close()
    if comm is not None and hasattr(comm, "close"):
        comm.close()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
