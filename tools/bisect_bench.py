"""Debug helper: run the bench workload stage by stage with progress prints (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from camkifu_amd import capi, synth
F = int(sys.argv[1]); H, W = 1080, 1920
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
rng = np.random.default_rng(1)
corners = synth.random_corners(H, W, rng)
frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
for i in range(F):
    if i < 4:
        frames[i] = synth.render(H, W, synth.random_stones(rng, 0.3), corners, seed=i, device=dev)
    else:
        frames[i] = frames[i % 4]
torch.cuda.synchronize(); print("rendered", flush=True)
def t(name, fn):
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-14s ok %.1f ms" % (name, 1e3 * (time.perf_counter() - t0)), flush=True); return r
e = t("board_edges", lambda: ctx.board_edges(frames))
out = t("board_lines", lambda: ctx.board_lines(e))
print("lines0", out[0]["n_lines"], out[0]["n_contours"], flush=True)
out = t("board_detect", lambda: ctx.board_detect(frames))
dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
M = capi.get_perspective_transform(corners, dst)
g = t("warp", lambda: ctx.warp_perspective(frames, M))
ctx.cnn_set_weights(synth.cnn_weights())
r = t("cnn", lambda: ctx.cnn_predict(g, want_y=False))
r = t("stones_detect", lambda: ctx.stones_detect(frames, M))
for k in range(2):
    t("board_detect#%d" % k, lambda: ctx.board_detect(frames))
    t("stones#%d" % k, lambda: ctx.stones_detect(frames, M))
print("done", flush=True)
