"""Loop trip counts of describe_tile_kernel on the bench's scenes (for profiles/valu_loop_weights.json): per wave of the launch, the mean
number of blur windows (window loop) and of keypoint groups (group loop), from the keypoints a device batch returns and the tiling
orb_prepare_geometry chooses (restated here: the cut with the fewest 48 x 48 windows, tiles of at most 153 x 153 keypoint positions).
  python scripts/describe_tile_stats.py [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import orb_slam3_rust_amd as P

NWX = NWY = 4
TW_MAX = 48 * NWX - 39; TH_MAX = 48 * NWY - 39

def cut(k, tmax, margin):
    best = None
    n0 = -(-k // tmax)
    for n in range(n0, n0 + 5):
        t = -(-k // n); wins = n * (-(-(t + margin) // 48))
        if best is None or wins < best[0]:
            best = (wins, n, t)
    return best[1], best[2]

def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    W, H, N = 752, 480, 2000
    cam = P.CameraModel(**P.synth.EUROC_CAMERA)
    h = P.Handle(cam, N, device=0, max_w=W, max_h=H, max_batch=pairs)
    imgs = np.stack([np.stack(P.synth.stereo_pair(1001, f)) for f in range(pairs)])
    o = h.alloc_batch_outputs(pairs, N + 64)
    h.process_stereo_batch_device(torch.from_numpy(imgs).cuda(), o)
    h.synchronize()
    kp = o["kp"].cpu().numpy().reshape(pairs * 2, -1, 7); nkp = o["nkp"].cpu().numpy().reshape(-1)
    sc = [float(np.float32(np.float64(np.float32(1.2)) ** l)) for l in range(8)]
    lv = [(int(np.rint(np.float32(W) / np.float32(s))), int(np.rint(np.float32(H) / np.float32(s)))) for s in sc]
    geo = []
    for (w, hh) in lv:
        nx, tw = cut(w - 62, TW_MAX, 39); ny, th = cut(hh - 62, TH_MAX, 39)
        geo.append((nx, ny, tw, th))
    win_trips = []; grp_trips = []; empty = 0; tiles = 0; windows = 0; touched = 0; kp_hist = []; chunk_rounds = np.zeros(4)
    for i in range(pairs * 2):
        k = kp[i, :nkp[i]]
        octv = np.ascontiguousarray(k).view(np.int32)[:, 5]
        for l, ((w, hh), (nx, ny, tw, th)) in enumerate(zip(lv, geo)):
            sel = k[octv == l]
            kx = np.rint(sel[:, 0] / np.float32(sc[l])).astype(int); ky = np.rint(sel[:, 1] / np.float32(sc[l])).astype(int)
            t = ((ky - 31) // th) * nx + (kx - 31) // tw
            cnt = np.bincount(t, minlength=nx * ny)
            for ty in range(ny):
                for tx in range(nx):
                    n = cnt[ty * nx + tx]; tiles += 1
                    if n == 0:
                        empty += 1; win_trips += [0] * 4; grp_trips += [0] * 4
                        continue
                    X0 = 31 + tx * tw; X1 = min(X0 + tw - 1, w - 32); Y0 = 31 + ty * th; Y1 = min(Y0 + th - 1, hh - 32)
                    pitch = w if l == 0 else (w + 63) // 64 * 64
                    ox = min(X0 - 21, pitch - 64) & ~3
                    oy = min(Y0 - 21, hh - 64); oy -= (4 - ((hh - 64 - oy) & 3)) & 3
                    nwin = min((X1 + 18 - (ox + 3) + 1 + 47) // 48, NWX) * min((Y1 + 18 - (oy + 3) + 1 + 47) // 48, NWY)
                    windows += nwin
                    # windows some keypoint's 37 x 37 patch reaches (blurred pixel (x, y) is in window ((x - ox - 3) // 48, (y - oy - 3) // 48))
                    nwx_ = min((X1 + 18 - (ox + 3) + 1 + 47) // 48, NWX); nwy_ = nwin // nwx_
                    m = (t == ty * nx + tx); hit = np.zeros((nwy_, nwx_), bool)
                    for x, y in zip(kx[m], ky[m]):
                        hit[max((y - 18 - oy - 3) // 48, 0):min((y + 18 - oy - 3) // 48, nwy_ - 1) + 1, max((x - 18 - ox - 3) // 48, 0):min((x + 18 - ox - 3) // 48, nwx_ - 1) + 1] = True
                    touched += int(hit.sum()); kp_hist.append(n)
                    g = (n + 3) // 4
                    for wv in range(4):
                        win_trips.append(len(range(wv, nwin, 4))); grp_trips.append(len(range(wv, g, 4)))
                        # round 6 form: a wave's groups in chunks of four rounds; chunk_rounds[j] counts the chunks that hold a round j
                        R = len(range(wv, g, 4))
                        for c0 in range(0, R, 4):
                            for j in range(min(4, R - c0)):
                                chunk_rounds[j] += 1
    print("images %d, tiles per image %.1f, empty tiles %.3f, windows per image %.1f, keypoints per image %.1f" % (pairs * 2, tiles / (pairs * 2), empty / tiles, windows / (pairs * 2), nkp.mean()))
    print("window loop trips per wave (all waves of the launch) %.4f; keypoint-group loop trips per wave %.4f; group passes per image %.1f" % (np.mean(win_trips), np.mean(grp_trips), np.sum(grp_trips) / (pairs * 2)))
    print("windows reached by some keypoint patch: %.1f per image (%.3f of the windows); keypoints per tile: mean %.1f, median %d, p90 %d, max %d" % (touched / (pairs * 2), touched / windows, np.mean(kp_hist), np.median(kp_hist), np.percentile(kp_hist, 90), np.max(kp_hist)))
    nw = len(grp_trips)
    print("chunks of four rounds per wave (all waves of the launch): %.4f; of which hold a round 1 / 2 / 3: %.4f / %.4f / %.4f" % (chunk_rounds[0] / nw, chunk_rounds[1] / chunk_rounds[0], chunk_rounds[2] / chunk_rounds[0], chunk_rounds[3] / chunk_rounds[0]))
    h.close()

main()
