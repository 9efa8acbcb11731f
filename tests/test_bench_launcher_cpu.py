"""`python bench.py --gpus N` from the PLAIN command (no torch.distributed.run): bench.py starts its N ranks itself.
The launcher is tested alone here (gloo, no GPU): --launch-check makes every rank join a group and add up the ranks."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(extra), env=env, capture_output=True,
                       text=True, timeout=timeout)
    return p, time.time() - t0


def test_plain_command_starts_its_ranks_and_relays_one_line():
    p, _ = _run("--gpus", "2", "--launch-check", "ok")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                       # stdout is rank 0's ONE line; the other ranks talk on stderr
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "world": 2, "local_rank": 0, "sum": 3.0}
    assert "rank 1 of 2 is up (LOCAL_RANK 1" in p.stderr


def test_four_ranks_get_their_own_local_rank():
    p, _ = _run("--gpus", "4", "--launch-check", "ok")
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["sum"] == 10.0
    for r in (1, 2, 3):                                    # every child got its own RANK and LOCAL_RANK (= its GPU), loopback rendezvous
        assert "rank %d of 4 is up (LOCAL_RANK %d, MASTER_ADDR 127.0.0.1)" % (r, r) in p.stderr


def test_a_rank_that_dies_fails_the_command_and_nobody_is_left_waiting():
    p, took = _run("--gpus", "3", "--launch-check", "fail:1")
    assert p.returncode == 3
    assert "rank 1 exited with code 3" in p.stderr
    assert took < 120                                      # the survivors sat in the rendezvous: killed, not waited for


def test_under_an_external_launcher_nothing_is_spawned():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check", "ok"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["world"] == 1


def test_a_rank_that_never_joins_ends_the_run_inside_the_limit_with_a_diagnosis():
    """a first-contact hang (a rank that never reaches the rendezvous, a communicator that never comes
    up) used to sit until somebody's outer limit with nothing to read.  Rank 2 of 3 sleeps instead of joining: the
    launcher sees no output from anybody for its silent limit (10 s here, 300 s by default), says which ranks were alive
    and what each wrote last, kills its children and exits with code 3."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["CK_LAUNCH_SILENT_LIMIT"] = "10"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check", "hang:2"], env=env,
                       capture_output=True, text=True, timeout=180)
    took = time.time() - t0
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert took < 90
    assert p.stdout.strip() == ""                          # no bench line
    assert "no rank has written anything for 10 s" in p.stderr and "Ranks still alive: [0, 1, 2]" in p.stderr
    assert "rank 2 (alive) last wrote: rank 2 sleeps instead of joining the group" in p.stderr
    assert "rank 0 (alive) last wrote:" in p.stderr
