"""Latency of a SMALL board_detect call (what the hold-off-aware fold issues: a window of ~16 frames on a high-priority
context) on an idle GPU and next to the stones path running flat out on two other contexts; with CK_PROFILE_HOST=1 the
library prints its host-side lap times (each lap ends with a stream synchronisation), summed here per lap.
usage: python tools/board_call_latency.py [frames per call]"""
import collections
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np
    import torch
    from camkifu_amd import capi, pipeline, synth
    from camkifu_amd.stone.nn_manager import NNManager
    n = int(sys.argv[1])
    dev = torch.device("cuda:0")
    frames, corners = synth.film(128, 1080, 1920, seed=synth.SEED, device=dev, quiet=52, move_every=32, hand_frames=12)[:2]
    torch.cuda.synchronize()
    M = capi.get_perspective_transform(corners, np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
    ctx = capi.Context(0, priority=1)
    ctx.board_detect(frames[:n], -1, pipeline.LMAX, True)
    stop = threading.Event()

    def load():
        torch.cuda.set_device(0)
        c = capi.Context(0)
        c.cnn_set_weights(NNManager.init_net())
        while not stop.is_set():
            c.stones_run(frames, M)
        c.close()
    for busy in (False, True):
        stop.clear()
        workers = [threading.Thread(target=load) for _ in range(2)] if busy else []
        for t in workers:
            t.start()
        time.sleep(1.0 if busy else 0.0)
        print("PHASE %s" % ("busy" if busy else "idle"), file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.board_detect(frames[:n], -1, pipeline.LMAX, True)
        dt = time.perf_counter() - t0
        print("PHASE end %.3f" % (1e3 * dt / 20), file=sys.stderr, flush=True)
        stop.set()
        for t in workers:
            t.join()
    sys.exit(0)

n = sys.argv[1] if len(sys.argv) > 1 else "16"
env = dict(os.environ, CK_PROFILE_HOST="1")
out = subprocess.run([sys.executable, __file__, n, "child"], env=env, stderr=subprocess.PIPE, text=True).stderr
phase, laps = None, {}
for line in out.splitlines():
    m = re.match(r"PHASE (\w+)(?: ([\d.]+))?", line)
    if m:
        if m.group(1) == "end":
            tot = sum(laps[phase].values()) / 20
            print("%s: %.2f ms per call of %s frames; inside k_board_lines %.2f ms:" % (phase, float(m.group(2)), n, tot))
            for k, v in laps[phase].items():
                print("    %-20s %.3f ms" % (k, v / 20))
            phase = None
        else:
            phase = m.group(1)
            laps[phase] = collections.OrderedDict()
        continue
    m = re.match(r"\[board_lines\] (.+?)\s+([\d.]+) ms", line)
    if m and phase:
        laps[phase][m.group(1)] = laps[phase].get(m.group(1), 0.0) + float(m.group(2))
