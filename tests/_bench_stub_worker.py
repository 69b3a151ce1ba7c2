"""Stand-in for bench.py's GPU worker (CUDAMAT_BENCH_WORKER_CMD): lets the CPU suite drive the launcher and the
per-form supervision of bench.py without a GPU.  Behaviour by exchange form (CUDAMAT_BENCH_FORMS):
  hang:*   every rank sleeps for ever                      (a collective that never completes)
  die:*    rank 1 exits with code 3, the others sleep      (one rank fails, its peers sit in a collective)
  slow:*   as a good form, after 2 s
  other    rank 0 prints one JSON line, everybody exits 0
"""
import json
import os
import sys
import time

form = os.environ["CUDAMAT_BENCH_FORMS"]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ.get("CUDAMAT_BENCH_WORKER") == "1" and os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "False"
print("stub chatter that is not the result line")
kind = form.split(":")[0]
if kind == "hang":
    time.sleep(1e6)
if kind == "die":
    if rank == 1:
        sys.exit(3)
    time.sleep(1e6)
if kind == "slow":
    time.sleep(2)
if rank == 0:
    print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "argv": sys.argv[1:], "master_port": os.environ["MASTER_PORT"],
                      "comm": {"gate": [{"form": form, "failed_ranks": 0}]}}), flush=True)
