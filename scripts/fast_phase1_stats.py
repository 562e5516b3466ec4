"""Statistics of fast_kernel's phase 1 on the synthetic scenes, for profiles/valu_loop_weights.json (numpy only, no GPU, no library):
tiles per image, 16-position tasks per tile, share of positions that pass the compass pre-test, and the mean over a tile's four waves of
the largest per-lane pass count (the trip count of the list append's bit walk).  The pyramid is a plain bilinear one — statistics, not parity.
usage: python scripts/fast_phase1_stats.py [pairs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_rust_amd as P  # noqa: E402  (synth only: nothing here touches the GPU)

EDGE, FT, T, LEVELS, SF = 31, 62, 20, 8, 1.2


def resize(img, w, h):
    H, W = img.shape
    fx = (np.arange(w) + 0.5) * (W / w) - 0.5
    fy = (np.arange(h) + 0.5) * (H / h) - 0.5
    x0 = np.clip(np.floor(fx).astype(int), 0, W - 2); y0 = np.clip(np.floor(fy).astype(int), 0, H - 2)
    ax = np.clip(fx - x0, 0, 1)[None, :]; ay = np.clip(fy - y0, 0, 1)[:, None]
    a = img[y0][:, x0] * (1 - ax) + img[y0][:, x0 + 1] * ax
    b = img[y0 + 1][:, x0] * (1 - ax) + img[y0 + 1][:, x0 + 1] * ax
    return np.rint(a * (1 - ay) + b * ay)


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n_tiles = n_img = 0
    tasks = passed = positions = both = 0
    walk = rounds_any = waves = 0
    for f in range(pairs):
        for img in P.synth.stereo_pair(4242, f):
            n_img += 1
            lvl = img.astype(np.float64)
            for l in range(LEVELS):
                s = SF ** l
                w, h = int(round(752 / s)), int(round(480 / s))
                if l:
                    lvl = resize(lvl, w, h)
                im = lvl.astype(np.int32)
                v = im[3:-3, 3:-3]; r0 = im[6:, 3:-3]; r8 = im[:-6, 3:-3]; r4 = im[3:-3, 6:]; r12 = im[3:-3, :-6]
                br = ((r0 > v + T) | (r8 > v + T)) & ((r4 > v + T) | (r12 > v + T))
                dk = ((r0 < v - T) | (r8 < v - T)) & ((r4 < v - T) | (r12 < v - T))
                ps = np.zeros(im.shape, bool); ps[3:-3, 3:-3] = br | dk
                bo = np.zeros(im.shape, bool); bo[3:-3, 3:-3] = br & dk
                for y0 in range(EDGE, h - EDGE, FT):
                    for x0 in range(EDGE, w - EDGE, FT):
                        aw = min(FT, w - EDGE - x0) + 2; ah = min(FT, h - EDGE - y0) + 2
                        q16 = (aw + 15) >> 4
                        reg = np.zeros((ah, 16 * q16), bool)
                        src = ps[y0 - 1:y0 - 1 + ah, x0 - 1:x0 - 1 + 16 * q16]
                        reg[:src.shape[0], :src.shape[1]] = src
                        cnt = reg.reshape(ah, q16, 16).sum(2).reshape(-1)          # per task, task = j * q16 + g
                        lanes = np.zeros(256, int); lanes[:len(cnt)] = cnt
                        mx = lanes.reshape(4, 64).max(1)
                        n_tiles += 1; tasks += len(cnt); passed += int(reg.sum()); positions += reg.size
                        both += int(bo[y0 - 1:y0 - 1 + ah, x0 - 1:x0 - 1 + 16 * q16].sum())
                        walk += int(mx.sum()); rounds_any += int((mx > 0).sum()); waves += 4
    print("images %d  tiles per image %.1f  16-position tasks per tile %.1f  positions per tile %.1f" % (n_img, n_tiles / n_img, tasks / n_tiles, positions / n_tiles))
    print("pass the pre-test: %.1f per tile (%.2f %% of the positions); both polarities %.2f %% of the survivors" % (passed / n_tiles, 100.0 * passed / positions, 100.0 * both / max(passed, 1)))
    print("bit walk: mean over a tile's four waves of the largest per-lane count %.3f (waves with any pass: %.3f of them, mean among those %.3f)" %
          (walk / waves, rounds_any / waves, walk / max(rounds_any, 1)))


if __name__ == "__main__":
    main()
