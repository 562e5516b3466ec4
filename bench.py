#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: stereo frames/s of ORB extract + match + triangulate
at 752x480, 2000 ORB/frame (configs[1]), plus local-BA LM iterations/s (configs[2]) as extra keys.

    python bench.py --gpus N --steps K --warmup W [--batch B]

A "step" is one pass of the hot path (orbx_process_stereo_batch_device) over one batch of B synthetic
stereo pairs that are already resident in HBM.  For N > 1 the driver launches one rank per GPU
(torch.distributed.run); frames shard across ranks with no data-path collective, so
per-GPU work is fixed ("weak" scaling) and value = all ranks' frames / max-over-ranks time.

Prints ONE JSON line on rank 0 with the `roofline` and `cpu_baseline` objects described in DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# compulsory bytes per stereo frame at 752x480, N=2000
def algo_bytes_per_frame(w, h, n):
    return 2 * w * h + 2 * n * (28 + 32) + 2 * n * (32 + 8) + n * 16 + n * 25
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_batches(P, torch, dev, seed, batch, n_batches, w, h):
    """n_batches distinct [batch,2,h,w] u8 device tensors.  32 generated pairs are expanded on the GPU
    by vertical flips (rows stay rectified) and brightness offsets, so that the working set
    (n_batches*batch*2*w*h bytes) exceeds the 256 MiB Infinity Cache."""
    uniq = min(32, batch * n_batches)
    base = torch.from_numpy(P.synth.stereo_batch(seed, 0, uniq, w, h)).to(dev)
    out = []
    k = 0
    for _ in range(n_batches):
        items = []
        for _ in range(batch):
            img = base[k % uniq]
            v = k // uniq
            if v & 1:
                img = torch.flip(img, dims=[1])
            off = ((v >> 1) % 5) * 3 - 6
            if off:
                img = (img.to(torch.int16) + off).clamp_(0, 255).to(torch.uint8)
            items.append(img)
            k += 1
        out.append(torch.stack(items).contiguous())
    return out


def cpu_baseline(P, frames_mt, frames_1t, w, h, n_features):
    """The CPU oracle (a restatement of the reference algorithm, kind 'port') on a bounded sample of the
    same workload, timed on this box's host cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    flags = "-O2 (portable build)"
    try:
        O.use_native_build()          # g++ -O3 -march=native on this box
        flags = "-O3 -march=native"
    except Exception:
        O.lib()
    cam = O.Camera(**P.synth.EUROC_CAMERA)
    p = O.orb_params(n_features)
    imgs = P.synth.stereo_batch(999, 0, max(frames_mt, frames_1t), w, h)

    def one(b):
        kl, dl = O.orb_extract(imgs[b, 0], p)
        kr, dr = O.orb_extract(imgs[b, 1], p)
        return len(O.stereo_match(cam, kl, dl, kr, dr)[0])

    one(0)
    t0 = time.perf_counter()
    for b in range(frames_1t):
        one(b)
    t1 = time.perf_counter() - t0
    cores = min(os.cpu_count() or 1, 16)
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    t0 = time.perf_counter()
    passes = 3                                 # ~15 s of CPU work in total at 752x480 (bounded sample)
    with ThreadPoolExecutor(cores) as ex:      # ctypes releases the GIL inside the oracle
        list(ex.map(one, [b % frames_mt for b in range(passes * frames_mt)]))
    tm = time.perf_counter() - t0
    return dict(value=round(passes * frames_mt / tm, 3), unit="stereo frames/s", cores=cores, kind="port",
                sample="%d synthetic %dx%d stereo frames x %d passes, N=%d, oracle extract+match+triangulate (g++ %s), %d threads "
                       "(frame-level parallelism); 1 thread: %.3f frames/s on %d frames"
                       % (frames_mt, w, h, passes, n_features, flags, cores, frames_1t / t1, frames_1t),
                value_1thread=round(frames_1t / t1, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="stereo pairs per step per GPU")
    ap.add_argument("--n-batches", type=int, default=3)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-files", action="store_true", help="skip the EuRoC-directory (PNG decode inclusive) leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import orb_slam3_rust_amd as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ORBX_DIST_REHEARSE=1: rehearsal of the N>1 code path on a box with fewer GPUs than ranks (ranks share the
    # cards, gloo instead of RCCL, which refuses two ranks on one device).  Never set by the driver; the line it
    # prints says so in config.parallelism.
    rehearse = world > 1 and os.environ.get("ORBX_DIST_REHEARSE") == "1"
    if rehearse:
        local_rank %= torch.cuda.device_count()
    if world > 1:
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    W, H = args.width, args.height     # default = BASELINE configs[1]; 1920x1080 --features 4000 = configs[4]
    cam = P.CameraModel(**P.synth.EUROC_CAMERA)
    h = P.Handle(cam, args.features, device=local_rank, max_w=W, max_h=H, max_batch=args.batch)
    cap = args.features + 304
    out = h.alloc_batch_outputs(args.batch, cap)
    # rank r owns stream r (seed 1000*r+1): frames shard across ranks, no collective
    batches = make_batches(P, torch, dev, 1000 * P.dist.shard_streams(world, rank, world)[0] + 1, args.batch, args.n_batches, W, H)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        h.process_stereo_batch_device(batches[i % len(batches)], out)
    h.check_status()
    h.set_profiling(True)     # HIP events around every launch, on the library's stream
    barrier(); torch.cuda.synchronize(); h.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        h.process_stereo_batch_device(batches[i % len(batches)], out)
    h.synchronize(); torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    acc = {k: [v[0], v[1]] for k, v in h.kernel_times().items()}   # durations over the timed region
    h.set_profiling(False)
    h.check_status()
    # the same K steps once more without the per-kernel HIP events: reported as value_unprofiled, never as value
    barrier(); torch.cuda.synchronize(); h.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        h.process_stereo_batch_device(batches[i % len(batches)], out)
    h.synchronize(); torch.cuda.synchronize(); barrier()
    elapsed_np = time.perf_counter() - t1
    if world > 1:
        elapsed = P.dist.allreduce_max_seconds(elapsed, dev)
        elapsed_np = P.dist.allreduce_max_seconds(elapsed_np, dev)
    n_matches = float(out["nmatches"].float().mean().item())
    n_kp = float(out["nkp"].float().mean().item())

    # roofline of the dominant kernel (largest summed HIP-event duration over the timed region)
    total_ms = sum(a[0] for a in acc.values())
    dom = max(acc.items(), key=lambda kv: kv[1][0])
    dom_name, (dom_ms, dom_launches) = dom
    # algorithmic bytes per launch of each kernel: per image, x images per launch
    n_img = 2 * args.batch
    sc = [float(np.float32(np.float64(np.float32(1.2)) ** l)) for l in range(8)]       # level sizes as orb.cpp
    lv = [(int(np.rint(np.float32(W) / np.float32(s))), int(np.rint(np.float32(H) / np.float32(s)))) for s in sc]
    px = [a * b for a, b in lv]
    per_launch = {
        "fast_kernel": n_img * (sum(px) + 4 * 2 * args.features),          # read every level once, write candidates
        "blur_kernel": n_img * 2 * sum(px),                                # read + write every level
        "resize_kernel": n_img * (sum(px[:-1]) + sum(px[1:])) / 7.0,       # 7 launches: read l-1, write l
        "describe_kernel": n_img * args.features * (31 * 31 + 37 * 37 + 28 + 32),
        "harris_select_kernel": n_img * 2 * args.features * (81 + 4 + 8),
        "rank_select_kernel": n_img * 2 * args.features * 16,
        "stereo_bucket_kernel": args.batch * args.features * (8 + 12),
        "stereo_match_kernel": args.batch * (2 * args.features * (32 + 8) + args.features * 8),
        "stereo_compact_kernel": args.batch * args.features * (8 + 16 + 25),
    }
    ach = per_launch.get(dom_name, 0) / (dom_ms / max(dom_launches, 1) * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, run
    # separately: scripts/pmc_summary.py -> profiles/*_pmc_traffic.json); counters cannot be read live here
    traffic = None
    valu = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pmc.get("batch") == args.batch and pmc.get("n_features") == args.features:
            kd = pmc["kernels"].get(dom_name, {})
            traffic = kd.get("hbm_bytes_per_launch")
            # what actually binds these byte/integer kernels: VALU issue.  Wave-instructions per launch from the SQ pass
            # (same file), priced with the launch time measured live: a SIMD issues one wave64 VALU instruction per 4
            # cycles, 1024 SIMDs at 2.4 GHz = 614.4 G wave-instructions/s.
            if kd.get("valu_wave_instr_per_launch") and dom_ms > 0:
                rate = kd["valu_wave_instr_per_launch"] / (dom_ms / max(dom_launches, 1) * 1e-3) / 1e9
                valu = dict(wave_instr_per_launch=kd["valu_wave_instr_per_launch"], achieved=round(rate, 1), peak=614.4,
                            unit="G wave-instr/s", frac=round(rate / 614.4, 4))
    except Exception:
        traffic = None
    roofline = dict(bound="hbm", kernel=dom_name, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(ach / HBM_PEAK_GBS, 5), traffic=traffic, valu_issue=valu,
                    algorithmic_bytes_per_launch=int(per_launch.get(dom_name, 0)),
                    avg_launch_us=round(dom_ms / max(dom_launches, 1) * 1e3, 2),
                    kernel_ms_per_step={k: round(v[0] / args.steps, 4) for k, v in sorted(acc.items())},
                    path_algorithmic_GBps=round(algo_bytes_per_frame(W, H, args.features) * args.batch * world * args.steps / elapsed / 1e9, 3))

    frames = args.batch * args.steps * world
    value = frames / elapsed
    res = dict(metric="stereo frames/sec ORB extract+match @752x480", value=round(value, 2), unit="stereo frames/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8", data="synthetic",
               config=dict(workload="Synthetic %dx%d stereo, %d ORB/frame, extract+match+triangulate (BASELINE configs[%d])"
                                    % (W, H, args.features, 1 if (W, H, args.features) == (752, 480, 2000) else 4),
                           image=[W, H], n_features=args.features, batch_pairs_per_gpu=args.batch,
                           distinct_batches=args.n_batches, parallelism="frames sharded, %d rank(s), no collective%s" % (world, " (REHEARSAL: ranks share GPUs, gloo)" if rehearse else ""),
                           mean_keypoints_per_image=round(n_kp, 1), mean_matches_per_frame=round(n_matches, 1)),
               roofline=roofline, value_unprofiled=round(args.batch * args.steps * world / elapsed_np, 2))

    if not args.no_ba:
        try:
            res["local_ba"] = bench_ba(P, h, cam, rank, world, dev)
        except Exception as e:  # BA leg must not hide the headline number
            res["local_ba"] = dict(error=repr(e))
    if rank == 0 and world == 1 and not args.no_files and (W, H) == (752, 480):
        try:
            res["from_png_files"] = bench_from_files(P, h, torch, args.features)
        except Exception as e:
            res["from_png_files"] = dict(error=repr(e))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        big = W * H > 752 * 480
        res["cpu_baseline"] = cpu_baseline(P, 16 if big else 48, 2 if big else 12, W, H, args.features)
    elif rank == 0:
        res["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(res))
    h.close()
    if world > 1:
        dist.destroy_process_group()


def bench_from_files(P, h, torch, n_features, n_distinct=12, repeats=32, chunk=16):
    """The same path fed the way the reference is fed (io/euroc.rs:100-132): a synthetic EuRoC mav0 directory of PNG
    files -> orbx_euroc_read_pairs (host threads, PNG decode into pinned memory) -> orbx_process_stereo_batch
    (pipelined H2D / kernels / D2H).  Decode of chunk k+1 overlaps the GPU work of chunk k.  Reported beside the
    headline value, never as it: this number is bounded by the host's PNG decode rate."""
    import shutil
    import tempfile
    import threading
    root = tempfile.mkdtemp(prefix="orbx_mav0_")
    try:
        pairs = [P.synth.stereo_pair(77, i) for i in range(n_distinct)]
        ts0 = 1403636579763555584
        rows = []
        blobs = [(P.synth.png_encode(l, filters="cycle"), P.synth.png_encode(r, filters=4)) for l, r in pairs]
        P.synth.write_euroc_mav0(root, 1, seed=77)                      # directory skeleton + sensor.yaml
        for c in (0, 1):
            for f in os.listdir(os.path.join(root, "cam%d" % c, "data")):
                os.remove(os.path.join(root, "cam%d" % c, "data", f))
        first = {}
        for i in range(n_distinct * repeats):
            ts = ts0 + 50000000 * i
            rows.append("%d,%d.png" % (ts, ts))
            for c in (0, 1):
                path = os.path.join(root, "cam%d" % c, "data", "%d.png" % ts)
                key = (i % n_distinct, c)
                if key in first:
                    os.link(first[key], path)          # the same 12 distinct pairs again: hard links, not copies
                else:
                    with open(path, "wb") as f:
                        f.write(blobs[i % n_distinct][c])
                    first[key] = path
        for c in (0, 1):
            with open(os.path.join(root, "cam%d" % c, "data.csv"), "w") as f:
                f.write("#timestamp [ns],filename\n" + "\n".join(rows) + "\n")
        ds = P.EurocDataset(root)
        n = len(ds)
        threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)))   # the box's CPU share is 16
        bufs = [torch.zeros((chunk, 2, ds.height, ds.width), dtype=torch.uint8).pin_memory() for _ in range(2)]
        out = P.Handle.alloc_host_outputs(chunk, n_features + 304)
        ds.read_pairs(0, chunk, out=bufs[0].numpy(), threads=threads)   # warm: page cache, first launch
        h.process_stereo_batch_host(bufs[0], out)
        def one_pass():
            t0 = time.perf_counter()
            ds.read_pairs(0, min(chunk, n), out=bufs[0].numpy(), threads=threads)
            for k, first_pair in enumerate(range(0, n, chunk)):
                cnt = min(chunk, n - first_pair)
                nxt = first_pair + chunk
                th = None
                if nxt < n:
                    th = threading.Thread(target=ds.read_pairs, args=(nxt, min(chunk, n - nxt)), kwargs=dict(out=bufs[(k + 1) % 2].numpy(), threads=threads))
                    th.start()
                h.process_stereo_batch_host(bufs[k % 2][:cnt], out)
                if th:
                    th.join()
            return time.perf_counter() - t0

        el = min(one_pass(), one_pass())                 # two passes over the 384 frames, the better one (host-thread noise)
        t1 = time.perf_counter()
        ds.read_pairs(0, min(chunk, n), out=bufs[0].numpy(), threads=threads)
        decode_s = (time.perf_counter() - t1) / min(chunk, n)
        return dict(value=round(n / el, 1), unit="stereo frames/s", frames=n, decode_threads=threads,
                    decode_only_frames_per_s=round(1.0 / decode_s, 1), chunk_pairs=chunk,
                    note="PNG files on disk -> features on the host; host PNG decode bound")
    finally:
        shutil.rmtree(root, ignore_errors=True)


def bench_ba(P, h, cam, rank=0, world=1, dev=None):
    """configs[2]: local BA, 20 keyframes / 2000 map points, LM iterations per second.  One GPU: the
    whole window on this GPU.  N GPUs (configs[3]): the SAME window with its map points partitioned
    over the ranks and the reduced normal equations all-reduced over RCCL every iteration."""
    win = P.synth.ba_window(42, 20, 2000, P.BA_OBS)
    cfg = P.LocalBAConfigLM()
    hook = P.dist.make_allreduce_hook(dev) if world > 1 else None

    def solve():
        if world > 1:
            return P.dist.ba_solve_partitioned(h, cam, cfg, win["poses_cw"], win["fixed_cw"], win["points"], win["obs"],
                                               rank, world, hook)
        return h.ba_solve_visual(cam, cfg, win["poses_cw"], win["fixed_cw"], win["points"], win["obs"])

    r = solve()
    reps = 5
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    its = 0
    for _ in range(reps):
        r = solve()
        its += r["iterations"]
    dt = time.perf_counter() - t0
    if world > 1:
        dt = P.dist.allreduce_max_seconds(dt, dev)
    out = dict(workload="synth_ba(seed=42, K=20, M=2000), %d observations%s" % (
                   len(win["obs"]), ", points partitioned over %d ranks + RCCL all-reduce" % world if world > 1 else ""),
               lm_iters_per_s=round(its / dt, 2), ms_per_solve=round(dt / reps * 1e3, 3),
               iterations=r["iterations"], initial_error_px=round(r["initial_error"], 4),
               final_error_px=round(r["final_error"], 4))
    if world == 1:
        # SURVEY §8e "many BA windows, independent units": a single window is latency-bound (short dependent launches), so
        # several windows solved at once — one handle (= one HIP stream + workspaces) per host thread — share the GPU
        import threading
        nwin = 8
        hs = [P.Handle(cam, 100, device=dev.index if dev is not None else 0) for _ in range(nwin)]
        wins = [P.synth.ba_window(100 + i, 20, 2000, P.BA_OBS) for i in range(nwin)]
        counts = [0] * nwin

        def work(i):
            for _ in range(reps):
                counts[i] += hs[i].ba_solve_visual(cam, cfg, wins[i]["poses_cw"], wins[i]["fixed_cw"], wins[i]["points"], wins[i]["obs"])["iterations"]

        for i in range(nwin):
            hs[i].ba_solve_visual(cam, cfg, wins[i]["poses_cw"], wins[i]["fixed_cw"], wins[i]["points"], wins[i]["obs"])
        th = [threading.Thread(target=work, args=(i,)) for i in range(nwin)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dtc = time.perf_counter() - t0
        for x in hs:
            x.close()
        out["concurrent_windows"] = dict(windows=nwin, lm_iters_per_s=round(sum(counts) / dtc, 2), note="8 independent windows, one handle and host thread each")
    else:
        # the other natural sharding: every stream has its own map, so rank r solves ITS OWN window — no
        # collective in the data path; aggregate LM iterations/s over the ranks
        import torch
        import torch.distributed as dist
        own = P.synth.ba_window(100 + rank, 20, 2000, P.BA_OBS)
        h.ba_solve_visual(cam, cfg, own["poses_cw"], own["fixed_cw"], own["points"], own["obs"])
        dist.barrier()
        t0 = time.perf_counter()
        n_it = 0
        for _ in range(reps):
            n_it += h.ba_solve_visual(cam, cfg, own["poses_cw"], own["fixed_cw"], own["points"], own["obs"])["iterations"]
        dto = P.dist.allreduce_max_seconds(time.perf_counter() - t0, dev)
        tot = torch.tensor([float(n_it)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        out["independent_windows"] = dict(windows=world, lm_iters_per_s=round(float(tot.item()) / dto, 2),
                                          note="one window per rank (its own map), no collective")
    return out


if __name__ == "__main__":
    main()
