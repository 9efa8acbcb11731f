"""Synthetic goban scenes: a wood-coloured 19x19 board with black/white discs
under a random homography, on a grey background with a gradient and Gaussian noise.

The reference ships no sample clip, so every test and benchmark input comes
from this seeded generator.  Rendering uses torch so it runs on the CPU (tests) or directly
in HBM (bench.py).  Colours follow SURVEY.md: wood BGR ~ (65,100,128), black ~25, white ~230.
"""
import math

import numpy as np
import torch

SEED = 20161001
GSIZE = 19
E, B, W = 0, 1, 2


def random_corners(h, w, rng, jitter=0.05, fill=0.84, slant=0.06, whole_degrees=True):
    """Four board corners (outer wood edge), clockwise from top-left, as float32 (4,2) (x,y).
    The quad covers well over a third of the frame (area gate, bf_auto.py:82).  Both vertical
    sides are slanted by at least `slant`/2 of the board side: the reference groups line
    intersections by their x coordinate only (bf_auto.py:161, imgutil.py:59), so a board whose
    left (or right) corners share an x can never be detected by it."""
    side = fill * min(h, w)
    cx, cy = w / 2.0, h / 2.0
    base = np.array([[cx - side / 2, cy - side / 2], [cx + side / 2, cy - side / 2],
                     [cx + side / 2, cy + side / 2], [cx - side / 2, cy + side / 2]], np.float64)
    base += rng.uniform(-jitter, jitter, (4, 2)) * side
    for top, bot in ((0, 3), (1, 2)):
        sgn = 1.0 if rng.random() < 0.5 else -1.0
        mid = 0.5 * (base[top, 0] + base[bot, 0])
        half = rng.uniform(0.5, 1.0) * slant * side * 0.5
        base[top, 0], base[bot, 0] = mid + sgn * half, mid - sgn * half
    if whole_degrees:
        base = _snap_sides(base)
    return base.astype(np.float32)


def _snap_sides(quad):
    """Turn every side about its midpoint onto the nearest whole degree.  cv2.HoughLines votes in
    1-degree x 1-pixel bins with a threshold of min(h, w) / 5 (bf_auto.py:128): a ~900 px board edge
    lying between two angle bins smears its votes over 5-8 rho bins and no bin reaches the
    threshold, so at 1080p the reference's finder only ever locks on when the edges sit close to
    whole degrees -- which is how this synthetic camera is placed."""
    lines = []
    for i in range(4):
        p, q = quad[i], quad[(i + 1) % 4]
        mid = 0.5 * (p + q)
        ang = np.degrees(np.arctan2(q[1] - p[1], q[0] - p[0]))
        snapped = np.round(ang)
        if i in (1, 3) and abs(abs(snapped) - 90.0) < 2.0:       # keep the vertical sides slanted by >= 2 degrees
            snapped = np.sign(snapped) * (90.0 + (2.0 if abs(ang) >= 90.0 else -2.0))
        d = np.array([np.cos(np.radians(snapped)), np.sin(np.radians(snapped))])
        lines.append((mid, d))
    out = np.empty((4, 2), np.float64)
    for i in range(4):
        (m0, d0), (m1, d1) = lines[i - 1], lines[i]               # corner i = side (i-1) meets side i
        t = np.linalg.solve(np.array([d0, -d1]).T, m1 - m0)
        out[i] = m0 + t[0] * d0
    return out


def random_stones(rng, density=0.3, keep_first_line_empty=True):
    g = np.zeros((GSIZE, GSIZE), np.uint8)
    m = rng.random((GSIZE, GSIZE)) < density
    g[m] = rng.integers(1, 3, m.sum())
    if keep_first_line_empty:          # documented limitation of the auto finder (bf_auto.py:14-15)
        g[0, :] = g[-1, :] = 0
        g[:, 0] = g[:, -1] = 0
    return g


def homography(src, dst):
    """3x3 float64 H with H @ (src,1) ~ (dst,1); plain numpy solve (renderer only)."""
    A, b = [], []
    for (sx, sy), (dx, dy) in zip(src, dst):
        A.append([sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx]); b.append(dx)
        A.append([0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy]); b.append(dy)
    x = np.linalg.solve(np.array(A, np.float64), np.array(b, np.float64))
    return np.append(x, 1.0).reshape(3, 3)


def natural_texture(h, w, seed=SEED, device="cpu", contrast=25.0):
    """A 1/f-spectrum luminance field -- structure at every scale, as wood grain, cloth or carpet show it (and as the flat
    grey table of the default scene does not): random phases under a 1/f amplitude, zero mean, scaled so that 2.5 sigma =
    `contrast` grey levels.  Generated on the CPU from the seed (the same field on every device) -> float32 (h, w) on `device`."""
    g = torch.Generator()
    g.manual_seed(int(seed))
    fy = torch.fft.fftfreq(h)[:, None]
    fx = torch.fft.rfftfreq(w)[None, :]
    f = torch.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    amp = 1.0 / f
    amp[0, 0] = 0.0
    phase = torch.rand((h, w // 2 + 1), generator=g) * (2.0 * math.pi)
    field = torch.fft.irfft2(torch.polar(amp, phase), s=(h, w))
    field = field * (contrast / 2.5 / float(field.std()))
    return field.float().to(device)


def render(h, w, stones, corners, seed=SEED, noise=3.0, device="cpu", hand=None, background=None):
    """-> uint8 tensor (h, w, 3) BGR on `device`.  hand = (row, col): a player's hand and forearm reaching from the
    bottom edge of the board over that intersection (skin-coloured, about three cells wide).  background: a float (h, w)
    luminance field added to the grey table around the board (natural_texture)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    # image pixel -> board coordinates in cell units; the wood spans [-0.5, 18.5]^2 so that the
    # canonical 380x380 image puts intersection i at pixel 10 + 20 i (stonesfinder.py:964-981)
    lo, hi = -0.5, GSIZE - 0.5
    Hm = homography(np.asarray(corners, np.float64), [(lo, lo), (hi, lo), (hi, hi), (lo, hi)])
    Ht = torch.tensor(Hm, dtype=torch.float64, device=dev)
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float64),
                            torch.arange(w, device=dev, dtype=torch.float64), indexing="ij")
    den = Ht[2, 0] * xs + Ht[2, 1] * ys + Ht[2, 2]
    u = ((Ht[0, 0] * xs + Ht[0, 1] * ys + Ht[0, 2]) / den).float()
    v = ((Ht[1, 0] * xs + Ht[1, 1] * ys + Ht[1, 2]) / den).float()
    inside = (u >= lo) & (u <= hi) & (v >= lo) & (v <= hi)

    img = torch.empty((h, w, 3), dtype=torch.float32, device=dev)
    grad = 90.0 + 14.0 * (xs / w - 0.5).float() + 10.0 * (ys / h - 0.5).float()
    if background is not None:
        grad = grad + background.to(dev)
    img[:] = grad[..., None]
    wood = torch.tensor([65.0, 100.0, 128.0], device=dev)
    shade = 1.0 + 0.05 * torch.sin(u * 0.9) * torch.cos(v * 0.7)
    img[inside] = (wood[None, :] * shade[inside][:, None])
    # grid lines, 1/12 of a cell wide
    ru, rv = torch.round(u), torch.round(v)
    on_grid = inside & (u >= -0.04) & (u <= GSIZE - 1 + 0.04) & (v >= -0.04) & (v <= GSIZE - 1 + 0.04)
    line = on_grid & (((u - ru).abs() < 1.0 / 24) | ((v - rv).abs() < 1.0 / 24))
    img[line] = 35.0
    # stones
    st = torch.as_tensor(np.asarray(stones, np.uint8), device=dev)
    iu = ru.clamp(0, GSIZE - 1).long()
    iv = rv.clamp(0, GSIZE - 1).long()
    d2 = (u - ru) ** 2 + (v - rv) ** 2
    disc = inside & (d2 < 0.47 ** 2) & (ru >= 0) & (ru <= GSIZE - 1) & (rv >= 0) & (rv <= GSIZE - 1)
    col = st[iv, iu]              # stones[row=v][col=u]
    img[disc & (col == B)] = 25.0
    img[disc & (col == W)] = 230.0
    if hand is not None:
        hr, hc = float(hand[0]), float(hand[1])
        palm = (u - hc) ** 2 + (v - hr) ** 2 < 1.7 ** 2
        arm = ((u - hc).abs() < 1.2) & (v > hr) & (v < GSIZE + 3.0)
        skin = torch.tensor([135.0, 160.0, 205.0], device=dev)
        img[palm | arm] = skin * (1.0 + 0.04 * torch.sin(3.0 * (u + v)))[palm | arm][:, None]
    if noise > 0:
        img += torch.randn((h, w, 3), generator=g, device=dev) * noise
    return img.clamp_(0, 255).round_().to(torch.uint8)


def scene(h, w, seed=SEED, density=0.3, device="cpu", noise=3.0):
    """One seeded scene -> dict(frame uint8 (h,w,3) tensor, corners float32 (4,2), stones uint8 (19,19))."""
    rng = np.random.default_rng(seed)
    corners = random_corners(h, w, rng)
    stones = random_stones(rng, density)
    frame = render(h, w, stones, corners, seed=seed, noise=noise, device=device)
    return dict(frame=frame, corners=corners, stones=stones)


def video(nframes, h, w, seed=SEED, device="cpu", new_stone_every=5, noise=3.0):
    """A fixed camera over a game: one new stone every `new_stone_every` frames.
    -> frames uint8 (n,h,w,3) tensor, corners, list of per-frame stone grids, move list."""
    rng = np.random.default_rng(seed)
    corners = random_corners(h, w, rng)
    stones = np.zeros((GSIZE, GSIZE), np.uint8)
    frames, grids, moves = [], [], []
    color = B
    for f in range(nframes):
        if f % new_stone_every == 0 and f > 0:
            while True:
                r, c = rng.integers(1, GSIZE - 1, 2)
                if stones[r, c] == E:
                    break
            stones[r, c] = color
            moves.append((color, int(r), int(c)))
            color = W if color == B else B
        frames.append(render(h, w, stones, corners, seed=seed + f, noise=noise, device=device))
        grids.append(stones.copy())
    return torch.stack(frames), corners, grids, moves


def film(nframes, h, w, seed=SEED, device="cpu", density=0.25, quiet=52, move_every=40, hand_frames=12, noise=3.0,
         select=None, background=None):
    """A fixed camera over a game in progress, with the players' hands: the position at the start holds random
    stones; after `quiet` frames a move is played every `move_every` frames -- a hand covers the point for
    `hand_frames` frames, and when it leaves the new stone is there.  That is what SfNeural's steady state needs to
    see a move (foreground agitation, then calm: sf_neural.py:72-154).
    `select`: frame numbers to render (a rank's shard of the film); the game is scripted for all `nframes` regardless.
    `background`: a luminance field for the table around the board (natural_texture), the same in every frame.
    -> frames uint8 (len(select) or n, h, w, 3) tensor on `device`, corners, truth (n,19,19) uint8 = stones actually on
    the board in each frame, moves [(color, r, c, frame at which the stone is first visible)], hands (n,) bool"""
    rng = np.random.default_rng(seed)
    corners = random_corners(h, w, rng)
    stones = random_stones(rng, density)
    wanted = {int(g): k for k, g in enumerate(range(nframes) if select is None else select)}
    frames = torch.empty((len(wanted), h, w, 3), dtype=torch.uint8, device=device)
    truth = np.zeros((nframes, GSIZE, GSIZE), np.uint8)
    hands = np.zeros(nframes, bool)
    moves, color, pending = [], B, None
    for f in range(nframes):
        k = f - quiet
        hand = None
        if k >= 0:
            phase = k % move_every
            if phase == 0:
                while True:
                    r, c = (int(v) for v in rng.integers(2, GSIZE - 2, 2))
                    if stones[r, c] == E:
                        break
                pending = (color, r, c)
                color = W if color == B else B
            if pending is not None and phase < hand_frames:
                hand = pending[1:]
            elif pending is not None:
                stones[pending[1], pending[2]] = pending[0]
                moves.append(pending + (f,))
                pending = None
        if f in wanted:
            frames[wanted[f]] = render(h, w, stones, corners, seed=seed * 31 + f, noise=noise, device=device, hand=hand,
                                       background=background)
        truth[f] = stones
        hands[f] = hand is not None
    return frames, corners, truth, moves, hands


def cnn_weights(seed=SEED, as_torch=False, device="cpu"):
    """He-normal synthetic weights in Keras-1 'tf' layout; conv1 is scaled by 1/128 because the
    reference feeds raw 0..255 pixels (nn_cache.py:47-51) and random weights have no reason to
    compensate for it.  Replaced by the trained model when camkifu_amd/data/keras.h5 exists (NNManager.init_net)."""
    from .capi import WEIGHT_SHAPES, WEIGHT_ORDER
    rng = np.random.default_rng(seed)
    Wt = {}
    for k in WEIGHT_ORDER:
        shp = WEIGHT_SHAPES[k]
        if k.endswith("b"):
            Wt[k] = (rng.standard_normal(shp) * 0.05).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[:-1]))
            Wt[k] = (rng.standard_normal(shp) * math.sqrt(2.0 / fan_in)).astype(np.float32)
    Wt["c1w"] *= np.float32(1.0 / 128)
    if as_torch:
        return {k: torch.from_numpy(v).to(device) for k, v in Wt.items()}
    return Wt


def bgr_to_i420(frame):
    """(h, w, 3) uint8 BGR -> flat I420 (h*w*3/2 bytes), BT.601 studio range, chroma averaged over
    2x2 blocks: how a synthetic clip is stored in a .y4m file (camkifu_amd.core.capture.write_y4m)."""
    f = np.asarray(frame).astype(np.int32)
    h, w = f.shape[:2]
    assert h % 2 == 0 and w % 2 == 0
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    y = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16
    u = ((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128
    v = ((112 * r - 94 * g - 18 * b + 128) >> 8) + 128

    def sub(c):
        return (c.reshape(h // 2, 2, w // 2, 2).sum(axis=(1, 3)) + 2) >> 2
    out = np.concatenate([y.reshape(-1), sub(u).reshape(-1), sub(v).reshape(-1)])
    return np.clip(out, 0, 255).astype(np.uint8)


def random_game(n_moves, rng, side=9, cool=0):
    """A legal game of `n_moves` alternating random moves played under the rules of Go (captures,
    no suicide, no ko retake) inside a side x side corner, so that groups do get captured.
    `cool`: nobody plays on a point for that many moves after a stone was captured there (players
    need the time to take the prisoners off the board).
    -> (moves [(color, r, c)], positions [n_moves] uint8 (19,19) AFTER each move, captured [per move list])"""
    from .golib_shim import B, W, Move, NP_TYPE, Rule, StateError, gsize
    rule = Rule()
    moves, positions, captured = [], [], []
    color = B
    while len(moves) < n_moves:
        hot = {(r, c) for caps in captured[len(captured) - cool:] for _, r, c in caps} if cool else set()
        for _ in range(200):
            r, c = int(rng.integers(0, side)), int(rng.integers(0, side))
            if (r, c) in hot:
                continue
            try:
                caps = rule.put(Move(NP_TYPE, (color, r, c)))
                break
            except StateError:
                continue
        else:
            break                                            # no legal move found: stop early
        moves.append((color, r, c))
        captured.append([(col, y, x) for col, x, y in caps])    # numpy (r, c)
        pos = np.zeros((gsize, gsize), np.uint8)
        for x in range(gsize):
            for y in range(gsize):
                if rule.stones[x][y] != 'E':
                    pos[y, x] = 1 if rule.stones[x][y] == B else 2
        positions.append(pos)
        color = W if color == B else B
    return moves, positions, captured
