"""bench.py --gpus N without a GPU in sight: the launcher (plain `python bench.py --gpus N`), the per-rank supervisors
under torch.distributed.run, and what they do when an exchange form hangs or a rank dies -- driven with a stand-in
worker (tests/_bench_stub_worker.py).  The real workers are covered by the -m gpu tests (tests/test_gpu_dist.py)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = "%s %s" % (sys.executable, os.path.join(ROOT, "tests", "_bench_stub_worker.py"))


def _run(forms, gpus=2, extra_env=None, timeout=300):
    env = dict(os.environ, CUDAMAT_BENCH_WORKER_CMD=STUB, CUDAMAT_BENCH_FORMS=forms, CUDAMAT_BENCH_FORM_TIMEOUT="6")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CUDAMAT_BENCH_WORKER"):
        env.pop(k, None)
    env.update(extra_env or {})
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return r, time.time() - t0


def test_plain_command_starts_its_own_ranks_and_relays_one_line():
    r, _ = _run("good:1")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert out["comm"]["launcher"] == [dict(out["comm"]["launcher"][0], form="good:1", ok=True, ranks=["ok", "ok"])]
    assert out["degraded_form"] is None


def test_a_form_that_hangs_is_replaced_by_fresh_processes():
    r, dt = _run("hang:1,good:0")
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    log = out["comm"]["launcher"]
    assert [e["form"] for e in log] == ["hang:1", "good:0"] and [e["ok"] for e in log] == [False, True]
    assert "time limit" in log[0]["ranks"] and log[0]["seconds"] >= 6
    assert out["comm"]["gate"][0]["form"] == "good:0"          # the line is the second form's
    assert dt < 120


def test_two_hanging_forms_leave_the_third_its_share_of_the_budget():
    """the driver ends a bench run after 600 s: the ladder keeps to ONE overall budget (CUDAMAT_BENCH_BUDGET, default 500 s
    from the start of bench.py), a form that hangs is cut where the later forms keep their reserve, and the line says at
    its top level that it was not timed on the first form"""
    r, dt = _run("hang:1,hang:0,good:0", extra_env={"CUDAMAT_BENCH_FORM_TIMEOUT": "", "CUDAMAT_BENCH_BUDGET": "90", "CUDAMAT_BENCH_RESERVE": "15"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    log = out["comm"]["launcher"]
    assert [e["ok"] for e in log] == [False, False, True], log
    assert log[0]["limit_s"] <= 60 and log[1]["limit_s"] <= log[0]["limit_s"] and log[2]["limit_s"] >= 5
    assert sum(e["seconds"] for e in log) < 90 and dt < 150
    assert "good:0" in out["degraded_form"] and "hang:1" in out["degraded_form"]
    for form in ("hang:1", "hang:0"):          # the launcher's log reaches stderr as it happens
        assert "form %s -> FAILED" % form in r.stderr


def test_default_limits_fit_the_drivers_time_limit():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert sum(b.FORM_LIMITS) <= 480 and b.BUDGET_S <= 540
    # every form hanging: the limits handed out never exceed the budget
    left, used = b.BUDGET_S, 0.0
    for i in range(3):
        lim = b._form_limit(i, 3, left)
        used, left = used + lim, left - lim
    assert used <= b.BUDGET_S


def test_a_rank_that_dies_ends_its_peers_within_seconds():
    r, _ = _run("die:1,slow:0", gpus=3, extra_env={"CUDAMAT_BENCH_FORM_TIMEOUT": "60"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    log = out["comm"]["launcher"]
    assert log[0]["ok"] is False and log[0]["ranks"][1] == "exit 3" and log[0]["ranks"][0] == "running"
    assert log[0]["seconds"] < 30          # not the 60 s limit: the exit code is noticed at the next poll
    assert log[1]["ok"] is True and out["n_gpus"] == 3


def test_every_form_failing_is_an_error_not_a_line():
    r, _ = _run("hang:1,die:0")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "every exchange form failed" in r.stderr


def test_eight_ranks():
    """the shape of the driver's scaling run: eight supervisors, eight workers, one line"""
    r, _ = _run("slow:1", gpus=8, extra_env={"CUDAMAT_BENCH_FORM_TIMEOUT": "120"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 8 and out["comm"]["launcher"][0]["ranks"] == ["ok"] * 8
