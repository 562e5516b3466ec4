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
VALU_PEAK_4CYC = 585.0  # G wave-instr/s, measured (profiles/r02_valu_issue_probe.txt): packed-i16 / perm / dot4 / cmp / ... class
VALU_PEAK_2CYC = 960.0  # same file: v_add_u32 / v_and_b32 / v_lshrrev_b32 / v_bitop3_b32 / f32 fma class


def two_cycle_share(kernel):
    """Share of a kernel's VALU instructions in the 2-cycle issue class: profiles/valu_class_mix.json, written by
    scripts/valu_class_mix.py from the disassembly of liborbx_hip.so's gfx950 code objects (opcode classes from the issue probes,
    loop trip counts stated in profiles/valu_loop_weights.json).  None when the file or the kernel is missing."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "valu_class_mix.json")))["kernels"][kernel]
        return d["share_2cycle"], d.get("share_2cycle_bounds"), d.get("model_vs_pmc")
    except Exception:
        return None, None, None


def mfma_share(kernel):
    """MFMA instructions as a share of the kernel's vector instructions (same file).  An MFMA holds its SIMD's vector issue port for 8 cycles
    (MI355X_MICROARCH.md, per-instruction constants) = two slots of the 4-cycle class."""
    try:
        w = json.load(open(os.path.join(ROOT, "profiles", "valu_class_mix.json")))["kernels"][kernel]["valu_weighted_per_wave"]
        return w.get("mfma", 0.0) / w["total"] if w.get("total") else 0.0
    except Exception:
        return 0.0


def make_batches(P, torch, dev, seed, batch, n_batches, w, h):
    """n_batches distinct [batch,2,h,w] u8 device tensors, EVERY pair unique (no flips / brightness copies of 32
    pairs): the scene generator of synth.stereo_pair (canvas 128, 600 rectangles + 300 discs per 752x480 of area, each at
    its own disparity 2..60 px, painted far to near, +-4 noise) run on the GPU with torch, one painting step for all pairs
    of a batch at a time.  Same statistics as the numpy generator the tests use, not the same bytes.  The working set
    (n_batches*batch*2*w*h bytes) exceeds the 256 MiB Infinity Cache at the default sizes."""
    out = []
    area = (w * h) / (752.0 * 480.0)
    n_rect, n_disc = int(600 * area), int(300 * area)
    n = n_rect + n_disc
    yy = torch.arange(h, device=dev, dtype=torch.int16).view(1, h, 1)
    xx = torch.arange(w, device=dev, dtype=torch.int16).view(1, 1, w)
    for bi in range(n_batches):
        g = torch.Generator(device=dev)
        g.manual_seed(0x5EED0000 + 7919 * seed + bi)
        B = batch
        disp = (torch.rand((B, n), generator=g, device=dev) * 58.0 + 2.0) * (w / 752.0)
        order = torch.argsort(disp, dim=1)                                     # far first: nearer objects occlude
        cx = (torch.rand((B, n), generator=g, device=dev) * w).to(torch.int16)
        cy = (torch.rand((B, n), generator=g, device=dev) * h).to(torch.int16)
        sw = torch.randint(6, 61, (B, n), generator=g, device=dev, dtype=torch.int16)
        sh = torch.randint(6, 61, (B, n), generator=g, device=dev, dtype=torch.int16)
        gray = torch.randint(0, 256, (B, n), generator=g, device=dev, dtype=torch.int16)
        disc = (torch.arange(n, device=dev) >= n_rect).view(1, n).expand(B, n)
        gat = lambda t: torch.gather(t, 1, order)
        cx, cy, sw, sh, gray, disc, dsp = gat(cx), gat(cy), gat(sw), gat(sh), gat(gray), gat(disc), gat(disp.round().to(torch.int16))
        left = torch.full((B, h, w), 128, dtype=torch.int16, device=dev)
        right = torch.full((B, h, w), 128, dtype=torch.int16, device=dev)
        for i in range(n):
            v = lambda t: t[:, i].view(B, 1, 1)
            x0 = v(cx) - v(sw) // 2; y0 = v(cy) - v(sh) // 2
            r = v(sw) // 2
            dy = yy - y0
            rows_rect = (dy >= 0) & (dy < v(sh))
            dyc = (dy - r).to(torch.int32)
            is_d = v(disc)
            g_i = v(gray)
            for img, xs in ((left, x0), (right, x0 - v(dsp))):
                dx = xx - xs
                m_rect = rows_rect & (dx >= 0) & (dx < v(sw))
                dxc = (dx - r).to(torch.int32)
                m_disc = (dyc * dyc + dxc * dxc) <= (r.to(torch.int32) * r.to(torch.int32))
                img.copy_(torch.where(torch.where(is_d, m_disc, m_rect), g_i, img))
        left += torch.randint(-4, 5, (B, h, w), generator=g, device=dev, dtype=torch.int16)
        right += torch.randint(-4, 5, (B, h, w), generator=g, device=dev, dtype=torch.int16)
        out.append(torch.stack([left.clamp_(0, 255).to(torch.uint8), right.clamp_(0, 255).to(torch.uint8)], 1).contiguous())
        del left, right
    return out


def make_batches_small(P, torch, dev, seed, batch, n_batches, w, h):
    """--small-gen (profiler passes only): 32 numpy-generated pairs expanded by flips and brightness offsets — the same kernels see
    the same kind of images, without the ~50 000 tiny torch dispatches of the GPU generator, which rocprofv3's counter collection
    does not survive on this stack (SIGSEGV inside a torch elementwise launch under --pmc)."""
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
    ap.add_argument("--steps", type=int, default=60)      # 60 x 2.3 ms: a timed region of ~0.14 s (20 steps = 45 ms was thin)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="stereo pairs per step per GPU (256 until round 4; round 5: 512 — same kernels, the launches' ramps and tails "
                                                          "over twice the work: +3.7 %, profiles/r05_batch_sweep.txt; the 256-pair figure stays in batch_sweep)")
    ap.add_argument("--n-batches", type=int, default=3)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-files", action="store_true", help="skip the EuRoC-directory (PNG decode inclusive) leg")
    ap.add_argument("--no-extras", action="store_true", help="skip batch sweep / PCIe-inclusive / latency / config0 legs")
    ap.add_argument("--small-gen", action="store_true", help="32 numpy pairs expanded instead of the GPU scene generator (no longer used by the profiler passes)")
    ap.add_argument("--save-batches", default=None, help="write the generated input batches (uint8 [n_batches, batch, 2, H, W]) to this .npy file")
    ap.add_argument("--load-batches", default=None, help="read the input batches from a file --save-batches wrote instead of generating them: the profiler's "
                                                         "counter passes run on the bench's own 768 unique pairs without the generator's ~50 000 torch launches in the profiled process")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import orb_slam3_rust_amd as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        # one process per GPU: the ranks are started by torch.distributed.run (the driver's contract); a bare
        # `python bench.py --gpus 8` would silently report a single-GPU figure
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; launch N>1 as python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                 "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ..." % (args.gpus, world))
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
    if args.load_batches:
        arr = np.load(args.load_batches, mmap_mode="r")
        if arr.shape != (args.n_batches, args.batch, 2, H, W) or arr.dtype != np.uint8:
            sys.exit("bench.py: %s holds %s %s, expected uint8 %s" % (args.load_batches, arr.dtype, arr.shape, (args.n_batches, args.batch, 2, H, W)))
        batches = [torch.from_numpy(np.ascontiguousarray(arr[i])).to(dev) for i in range(args.n_batches)]
    else:
        batches = (make_batches_small if args.small_gen else make_batches)(P, torch, dev, 1000 * P.dist.shard_streams(world, rank, world)[0] + 1,
                                                                            args.batch, args.n_batches, W, H)
    torch.cuda.synchronize()
    if args.save_batches and rank == 0:
        np.save(args.save_batches, np.stack([b.cpu().numpy() for b in batches]))

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        h.process_stereo_batch_device(batches[i % len(batches)], out)
    h.check_status()
    # Which kernel dominates, and the per-kernel table: an UNTIMED pass with HIP events around every launch (on the library's stream).
    # The timed region then brackets the dominant kernel's launches only — the roofline block needs that kernel's durations over the
    # timed region, and two event records per launch of every kernel cost the step about 1.5 % (value_unprofiled below is the same
    # K steps with no events at all).
    n_pre = max(3, min(args.steps, 10))
    h.set_profiling(True)
    h.synchronize()
    t_pre = time.perf_counter()
    for i in range(n_pre):
        h.process_stereo_batch_device(batches[i % len(batches)], out)
    h.synchronize()
    elapsed_pre = time.perf_counter() - t_pre            # rounds 1-3 measured `value` this way (events around every launch)
    acc_pre = {k: [v[0], v[1]] for k, v in h.kernel_times().items()}
    h.set_profiling(False)
    dom_pre = max(acc_pre.items(), key=lambda kv: kv[1][0])[0]
    h.set_profiling(True, only=dom_pre)
    barrier(); torch.cuda.synchronize(); h.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        h.process_stereo_batch_device(batches[i % len(batches)], out)
    h.synchronize(); torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    acc_dom = {k: [v[0], v[1]] for k, v in h.kernel_times().items()}   # the dominant kernel's durations over the timed region
    h.set_profiling(False)
    h.check_status()
    # per-kernel milliseconds scaled to the timed region's K steps: the dominant kernel from the timed region itself, the others from the untimed pass
    acc = {k: [v[0] * args.steps / n_pre, int(round(v[1] * args.steps / n_pre))] for k, v in acc_pre.items()}
    acc.update(acc_dom)
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
    # pixels of a level that a keypoint's 43-row x 48-byte window can reach (keypoints sit in [31, w-31) x [31, h-31): columns 8 .. w-7,
    # rows 10 .. h-11): what describe_fused_kernel must fetch from HBM ONCE per image — neighbouring keypoints' windows overlap 3.3-fold,
    # and all but the first read of a line come out of L2 (the XCD-aware block map keeps an image on one XCD)
    px_win = [max(0, a - 14) * max(0, b - 20) for a, b in lv]
    # describe_tile_kernel (round 5): a tile's blur windows cover the keypoint region + 21 px on every side — columns / rows 10 .. w-11 / h-11 of
    # a level, read ONCE per image from HBM (a tile's 36-px margin is read again by its neighbours, out of L2), and the centroid's 31-row discs
    # lie inside that rectangle
    px_tile = [max(0, a - 20) * max(0, b - 20) for a, b in lv]
    def _cut(k, tmax, margin):       # orb_prepare_geometry's cut of a level's keypoint region into describe tiles: (windows, tiles)
        best = None
        n0 = -(-k // tmax)
        for n in range(n0, n0 + 5):
            t = -(-k // n); wins = n * (-(-(t + margin) // 48))
            if best is None or wins < best[0]:
                best = (wins, n)
        return best
    dt_windows = sum(_cut(a - 62, 153, 39)[0] * _cut(b - 62, 153, 39)[0] for a, b in lv if a > 62 and b > 62)
    # HBM bytes each kernel must move per launch (compulsory: every input byte once, every output byte once), x images per launch
    per_launch = {
        "fast_kernel": n_img * (sum(px) + 4 * 2 * args.features),          # read every level once, write candidates
        "blur_kernel": n_img * 2 * sum(px),                                # read + write every level
        "resize_kernel": n_img * (sum(px[:-1]) + sum(px[1:])) / 7.0,       # 7 launches: read l-1, write l
        "describe_fused_kernel": n_img * (sum(px_win) + args.features * (28 + 32)),    # the reachable part of every level once, the keypoint records and descriptors out
        "describe_tile_kernel": n_img * (sum(px_tile) + args.features * (8 + 8 + 28 + 32)),   # the same rectangle once, list entry + response in, keypoint record + descriptor out
        "describe_kernel": n_img * (2 * sum(px_win) + args.features * (28 + 32)),      # (ORBX_DESC_UNFUSED=1: the level and the blurred level)
        "harris_select_kernel": n_img * 2 * args.features * (81 + 4 + 8),
        "rank_select_kernel": n_img * 2 * args.features * 16,
        "stereo_bucket_kernel": args.batch * args.features * (8 + 12),
        "stereo_match_kernel": args.batch * (2 * args.features * (32 + 8) + args.features * 8),
        "stereo_compact_kernel": args.batch * args.features * (8 + 16 + 25),
    }
    # what the kernel REQUESTS from the memory hierarchy (L2 and below), overlap counted every time — not an HBM figure, never `achieved`
    l2_requests = {
        "describe_fused_kernel": n_img * args.features * (43 * 43 + 28 + 32),
        "describe_tile_kernel": n_img * (dt_windows * 64 * 64 + args.features * (31 * 4 * 12 + 8 + 8 + 28 + 32)),   # 64 x 64 bytes per blur window, 4 x 12 bytes per centroid row
        "describe_kernel": n_img * args.features * (31 * 31 + 37 * 37 + 28 + 32),
    }
    launch_s = dom_ms / max(dom_launches, 1) * 1e-3
    ach = per_launch.get(dom_name, 0) / launch_s / 1e9 if dom_ms > 0 else 0.0
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
            # (same file), priced with the launch time measured live against the issue rates MEASURED on this chip by
            # scripts/valu_issue_probe (profiles/r02_valu_issue_probe.txt): packed i16 / f16, v_perm_b32, v_alignbyte_b32,
            # v_dot4/dot2, 24-bit multiplies, v_cmp, v_cndmask, v_mbcnt, DPP moves, v_min/max_i32, shifts left issue once per
            # 4 cycles per SIMD at any occupancy (560-590 G wave-instr/s chip-wide: VALU_PEAK_4CYC); v_add/sub_u32,
            # v_and/or/xor, v_lshrrev, v_mov, v_bitop3, f32 add/mul/fma and v_min_u16 once per 2 (950-1080 G/s: VALU_PEAK_2CYC).
            # No hardware counter gives VALU busy time here — SQ_ACTIVE_INST_VALU and SQ_THREAD_CYCLES_VALU both count
            # instructions (profiles/r02_pmc_counter_calibration.txt) — so a kernel that mixes the classes is priced by its mix:
            # frac = rate x (share_2cycle / peak_2cycle + (1 - share_2cycle) / peak_4cycle), the share from scripts/valu_class_mix.py
            # (disassembly of the shipped code object, loop trip counts stated in profiles/valu_loop_weights.json).
            if kd.get("valu_wave_instr_per_launch") and dom_ms > 0:
                rate = kd["valu_wave_instr_per_launch"] / launch_s / 1e9
                share2, share2_bounds, model_vs_pmc = two_cycle_share(dom_name)
                sh = share2 if share2 is not None else 0.0
                ms_ = mfma_share(dom_name)                       # MFMAs among the counted vector instructions: 8 cycles of the issue port each
                slots = (1.0 - ms_) * (sh / VALU_PEAK_2CYC + (1.0 - sh) / VALU_PEAK_4CYC) + ms_ * 2.0 / VALU_PEAK_4CYC
                valu = dict(wave_instr_per_launch=kd["valu_wave_instr_per_launch"], achieved=round(rate, 1), peak=VALU_PEAK_4CYC,
                            peak_simple_ops=VALU_PEAK_2CYC, peak_for_this_mix=round(1.0 / slots, 1), unit="G wave-instr/s", two_cycle_share=share2, two_cycle_share_bounds=share2_bounds,
                            class_mix_model_vs_pmc=model_vs_pmc,
                            mfma_share=round(ms_, 4),
                            frac=round(rate * slots, 4),
                            frac_of_4cycle_class_rate=round(rate / VALU_PEAK_4CYC, 4),
                            source="scripts/valu_class_mix.py -> profiles/valu_class_mix.json (class mix from the code object); "
                                   "profiles/r02_valu_issue_probe.txt + profiles/r03_valu_issue_probe2.txt (class rates); profiles/pmc_traffic.json "
                                   "(instructions per launch); profiles/r02_pmc_counter_calibration.txt (what the counters count)",
                            counters_taken_on=pmc.get("input", "32 numpy pairs expanded (--small-gen)"), counters_unique_pairs=pmc.get("unique_pairs", 32),
                            lds_busy_frac_pmc=kd.get("lds_busy_frac"), lds_bank_conflict_share_pmc=kd.get("lds_bank_conflict_share"))
    except Exception:
        traffic = None
    # the >= 60 % HBM goal is assessed on the streaming kernels — algorithmic bytes / measured time / 8 TB/s each
    streaming = {}
    for kname in ("resize_kernel", "blur_kernel", "fast_kernel"):
        if kname in acc and acc[kname][0] > 0:
            per_step_bytes = per_launch[kname] * (7 if kname == "resize_kernel" else 1)
            gbs = per_step_bytes / (acc[kname][0] / args.steps * 1e-3) / 1e9
            streaming[kname.replace("_kernel", "")] = dict(GBps=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
    # The block names the limit that BINDS the dominant kernel.  hbm: compulsory bytes / launch time / 8 TB/s (cross-checked by the counter
    # bytes of the PMC pass, `traffic`).  When the kernel's VALU issue fraction (above) exceeds its HBM fraction the kernel is issue-bound and
    # achieved / peak / frac are the issue figures — wave-instructions per second against the rate this chip issues THIS kernel's opcode mix
    # at — with the HBM side kept beside them as `hbm` / `hbm_frac`.
    hbm_frac = ach / HBM_PEAK_GBS
    hbm = dict(achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(hbm_frac, 5), compulsory_bytes_per_launch=int(per_launch.get(dom_name, 0)),
               counter_bytes_per_launch=traffic,
               frac_from_counter_bytes=round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 5) if traffic and dom_ms > 0 else None,
               l2_request_bytes_per_launch=(int(l2_requests[dom_name]) if dom_name in l2_requests else None))
    issue_bound = bool(valu and valu["frac"] > hbm_frac)
    if issue_bound:
        head = dict(bound="valu_issue", achieved=valu["achieved"], peak=valu["peak_for_this_mix"], unit="G wave-instr/s", frac=valu["frac"])
    else:
        head = dict(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(hbm_frac, 5))
    roofline = dict(head, kernel=dom_name, traffic=traffic, hbm_frac=round(hbm_frac, 5), hbm=hbm, valu_issue=valu, streaming=streaming,
                    binding_limit=head["bound"],
                    algorithmic_bytes_per_launch=int(per_launch.get(dom_name, 0)),
                    avg_launch_us=round(launch_s * 1e6, 2),
                    kernel_ms_per_step={k: round(v[0] / args.steps, 4) for k, v in sorted(acc.items())},
                    kernel_ms_per_step_source="%s: HIP events over the timed region; the other kernels: an untimed pass of %d steps with events around every launch" % (dom_pre, n_pre),
                    path_algorithmic_GBps=round(algo_bytes_per_frame(W, H, args.features) * args.batch * world * args.steps / elapsed / 1e9, 3),
                    path_hbm_frac=round(algo_bytes_per_frame(W, H, args.features) * args.batch * world * args.steps / elapsed / 1e9 / HBM_PEAK_GBS / world, 5))

    frames = args.batch * args.steps * world
    value = frames / elapsed
    res = dict(metric="stereo frames/sec ORB extract+match @%dx%d" % (W, H), value=round(value, 2), unit="stereo frames/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8", data="synthetic",
               config=dict(workload="Synthetic %dx%d stereo, %d ORB/frame, extract+match+triangulate (BASELINE configs[%d])"
                                    % (W, H, args.features, 1 if (W, H, args.features) == (752, 480, 2000) else 4),
                           image=[W, H], n_features=args.features, batch_pairs_per_gpu=args.batch,
                           distinct_batches=args.n_batches, unique_pairs=(min(32, args.batch * args.n_batches) if args.small_gen else args.batch * args.n_batches),
                           parallelism="frames sharded, %d rank(s), no collective%s" % (world, " (REHEARSAL: ranks share GPUs, gloo)" if rehearse else ""),
                           mean_keypoints_per_image=round(n_kp, 1), mean_matches_per_frame=round(n_matches, 1)),
               roofline=roofline, value_unprofiled=round(args.batch * args.steps * world / elapsed_np, 2),
               value_all_kernel_events=dict(value=round(args.batch * n_pre / elapsed_pre, 2), steps=n_pre,
                                            note="rank 0's untimed pass with HIP events around EVERY launch: how `value` was measured until round 3 "
                                                 "(since round 4 the timed region brackets the dominant kernel only)"))

    if rank == 0 and world == 1 and not args.no_extras:
        try:
            res.update(bench_extras(P, h, torch, batches, out, args, W, H))
        except Exception as e:
            res["extras_error"] = repr(e)
    if not args.no_ba:
        try:
            res["local_ba"] = bench_ba(P, h, cam, rank, world, dev)
        except Exception as e:  # BA leg must not hide the headline number
            res["local_ba"] = dict(error=repr(e))
        if world > 1 and isinstance(res["local_ba"], dict) and res["local_ba"].get("transport"):
            res["config"]["parallelism"] += "; local BA: map points partitioned over the ranks, normal equations all-reduced — " + res["local_ba"]["transport"]
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


def bench_extras(P, h, torch, batches, out, args, W, H):
    """The rest of SURVEY §8(d)'s measurement list, driver-run: batch sweep (device-resident), the PCIe-inclusive rate of the
    host-buffer batch call, the latency of the one-pair drop-in call, and configs[0] (EuRoC-shaped stream, n_features = 1200,
    main.rs:53) on the CPU oracle with the GPU's results checked against it."""
    r = {}
    # batch sweep: pairs resident in HBM, no per-kernel events
    sweep = {}
    for b in (1, 8, 64, 256, 512):
        if b > args.batch:
            continue
        sub = [x[:b].contiguous() for x in batches]
        steps = max(5, min(200, 2048 // b))
        for i in range(3):
            h.process_stereo_batch_device(sub[i % len(sub)], out)
        h.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            h.process_stereo_batch_device(sub[i % len(sub)], out)
        h.synchronize()
        dt = time.perf_counter() - t0
        sweep[str(b)] = dict(frames_per_s=round(b * steps / dt, 1), ms_per_step=round(dt / steps * 1e3, 4))
    h.check_status()
    r["batch_sweep"] = sweep
    # PCIe-inclusive: the same pairs in pinned host memory -> orbx_process_stereo_batch (chunked, upload / kernels / download on
    # three streams) -> results in pinned host memory
    host_in = batches[0].cpu().pin_memory()
    hout = P.Handle.alloc_host_outputs(args.batch, out["cap_kp"])
    for _ in range(2):                       # (the second call still first-touches part of the pinned result buffers: 15.9 ms against 7.7 at 512 pairs)
        h.process_stereo_batch_host(host_in, hout)
    reps = 4
    per_call = []
    t0 = time.perf_counter()
    for _ in range(reps):
        t1 = time.perf_counter()
        h.process_stereo_batch_host(host_in, hout)
        per_call.append(round((time.perf_counter() - t1) * 1e3, 3))
    dt = time.perf_counter() - t0
    bytes_in = host_in.numel()
    bytes_out = sum(v.numel() * v.element_size() for k, v in hout.items() if hasattr(v, "numel"))
    r["pcie_inclusive"] = dict(value=round(args.batch * reps / dt, 1), unit="stereo frames/s", pairs_per_call=args.batch,
                               h2d_MB_per_call=round(bytes_in / 1e6, 1), d2h_MB_per_call=round(bytes_out / 1e6, 1), ms_per_call=per_call,
                               note="orbx_process_stereo_batch: host images in, host results out (pinned), H2D + kernels + D2H pipelined; never `value`")
    # the one-pair drop-in call (StereoProcessor::process): H2D of two images, ~20 kernels replayed from a hipGraph, one D2H block
    L = host_in[0, 0].numpy(); R = host_in[0, 1].numpy()
    for _ in range(5):
        h.process_stereo(L, R)
    lat = []
    for _ in range(60):
        t0 = time.perf_counter()
        h.process_stereo(L, R)
        lat.append(time.perf_counter() - t0)
    lat.sort()
    r["single_pair_latency_ms"] = dict(median=round(lat[len(lat) // 2] * 1e3, 4), p90=round(lat[int(len(lat) * 0.9)] * 1e3, 4),
                                       note="orbx_process_stereo, host buffers in and out, includes the python/ctypes call")
    # configs[0]: EuRoC-shaped (752x480, cam0 intrinsics, 20 Hz) at n_features = 1200 — the reference's own CPU-runnable case.
    # The CPU side is the oracle (the reference itself cannot be built here); the GPU handle's output on the same frames is
    # compared with it bit for bit.
    if (W, H) == (752, 480):
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        try:
            O.use_native_build()
        except Exception:
            O.lib()
        ocam = O.Camera(**P.synth.EUROC_CAMERA)
        p = O.orb_params(1200)
        frames = [P.synth.stereo_pair(4242, i) for i in range(16)]
        ts = [1403636579763555584 + 50000000 * i for i in range(len(frames))]          # 20 Hz, ns

        def one(i):
            kl, dl = O.orb_extract(frames[i][0], p); kr, dr = O.orb_extract(frames[i][1], p)
            return (kl, dl, kr, dr) + tuple(O.stereo_match(ocam, kl, dl, kr, dr))
        t0 = time.perf_counter()
        ref1 = [one(i) for i in range(4)]
        t1 = (time.perf_counter() - t0) / 4
        cores = min(os.cpu_count() or 1, 16)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            ref = list(ex.map(one, range(len(frames))))
        tm = (time.perf_counter() - t0) / len(frames)
        h12 = P.Handle(P.CameraModel(**P.synth.EUROC_CAMERA), 1200, device=h.device, max_w=752, max_h=480, max_batch=1)
        ok = 0
        t0 = time.perf_counter()
        got = [h12.process_stereo(f[0], f[1]) for f in frames]
        tg = (time.perf_counter() - t0) / len(frames)
        for g, o in zip(got, ref):
            same = all(np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes() for a, b in zip(g[:5], o[:5]))
            same = same and np.array_equal(g[6], o[6]) and np.array_equal(g[5][g[6] == 1], o[5][o[6] == 1])
            ok += bool(same)
        h12.close()
        r["config0_cpu_plumbing"] = dict(workload="EuRoC-shaped 752x480 stereo stream, 20 Hz timestamps %d..%d ns, n_features=1200 (main.rs:53), cam0 intrinsics" % (ts[0], ts[-1]),
                                         cpu_frames_per_s_1thread=round(1.0 / t1, 2), cpu_frames_per_s=round(1.0 / tm, 2), cpu_threads=cores,
                                         cpu_kind="port (oracle, the reference crate cannot be built here)",
                                         gpu_single_pair_frames_per_s=round(1.0 / tg, 1), frames=len(frames),
                                         gpu_equals_cpu_bit_for_bit="%d/%d frames" % (ok, len(frames)))
    return r


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


def schur_executed_flop(K_opt, M):
    """f64 flop the Schur launch EXECUTES for one window (upper 16x16 tiles of the padded 6K x 6K product, k dimension padded to the
    k-split plan of ba_solve_batch: BA_PPS_TARGET = 32 points per split, at most 128 splits): one v_mfma_f64_16x16x4_f64 = 2048 flop."""
    P_ = max(16, (6 * K_opt + 15) // 16 * 16)
    nt = P_ // 16
    ksplit = max(1, min(128, (M + 31) // 32))
    pps = max(8, ((M + ksplit - 1) // ksplit + 7) // 8 * 8)
    rows = 3 * pps * ksplit
    return nt * (nt + 1) // 2 * (rows // 4) * 2048.0


def bench_ba_inertial(P, h, cam, K=10, M=2000, seed=42):
    """solve_inertial_ba (local_inertial_ba.rs:1074-1275; LocalInertialBAConfig::window_size = 10): what the local mapper runs once the
    IMU is initialised.  15-d keyframe states, 9-d preintegration + 6-d bias-walk edges beside the reprojection residuals."""
    w = P.Handle.pack_ba_windows([P.synth.inertial_window(seed, K, M, P.BA_OBS)])[0]   # (observations in page-locked memory)
    icfg = P.LocalInertialBAConfig()
    args = (cam, icfg, w["poses_wc"], w["velocities"], w["biases"], w["fixed_cw"], w["points"], w["obs"], w["edge_kf"], w["preint"])
    r = h.ba_solve_inertial(*args)
    reps = 20
    t0 = time.perf_counter()
    its = 0
    for _ in range(reps):
        its += h.ba_solve_inertial(*args)["iterations"]
    dt = time.perf_counter() - t0
    h.set_profiling(True)
    h.ba_solve_inertial(*args)
    kt = h.kernel_times()
    h.set_profiling(False)
    n_it = max(r["iterations"], 1)
    return dict(workload="synth inertial window(seed=%d, K=%d, M=%d), %d observations, %d IMU edges (15-d states: reduced system n = %d)"
                         % (seed, K, M, len(w["obs"]), len(w["edge_kf"]), 15 * K),
                lm_iters_per_s=round(its / dt, 2), ms_per_solve=round(dt / reps * 1e3, 3), iterations=r["iterations"],
                kernel_ms_per_iteration={k: round(v[0] / n_it, 4) for k, v in sorted(kt.items()) if k.startswith("ba_")})


def bench_ba_config5(P, h, cam, cfg, K=50, M=8000, seed=43):
    # observations at keypoint precision (f32 widened, as the reference's are: local_ba_lm.rs:870-872) in page-locked memory, handed over in the
    # 16-byte form (orbx_ba_solve_visual_obs32: 3.2 MB up per solve instead of 6.5); the 32-byte form of the same window is timed beside it
    base = P.synth.keypoint_precision(P.synth.ba_window(seed, K, M, P.BA_OBS))
    win = P.Handle.pack_ba_windows([base], obs32=True)[0]
    win64 = P.Handle.pack_ba_windows([base])[0]
    args = (cam, cfg, win["poses_cw"], win["fixed_cw"], win["points"], win["obs"])
    args64 = (cam, cfg, win64["poses_cw"], win64["fixed_cw"], win64["points"], win64["obs"])
    r = h.ba_solve_visual(*args)
    reps = 20                                    # (5 until round 5: one 3 ms stall of the host inside 14 ms of timed solves moved the figure by a quarter)
    per = []
    t0 = time.perf_counter()
    its = 0
    for _ in range(reps):
        t1 = time.perf_counter()
        its += h.ba_solve_visual(*args)["iterations"]
        per.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    r64 = h.ba_solve_visual(*args64)
    t0 = time.perf_counter()
    its64 = 0
    for _ in range(reps):
        its64 += h.ba_solve_visual(*args64)["iterations"]
    dt64 = time.perf_counter() - t0
    h.set_profiling(True)
    h.ba_solve_visual(*args)
    kt = h.kernel_times()
    h.set_profiling(False)
    n_it = max(r["iterations"], 1)
    k_opt = len(win["poses_cw"])
    flop = schur_executed_flop(k_opt, M)
    fused = kt.get("ba_kf_schur_kernel", (0.0, 1))
    fused_ms = fused[0] / max(fused[1], 1)
    kernel_ms = sum(v[0] for k, v in kt.items() if k.startswith("ba_"))
    res = dict(workload="synth_ba(seed=%d, K=%d, M=%d) at keypoint precision, %d observations (16-byte form, %.1f MB up per solve), %d optimised keyframes (reduced system n = %d, one-launch Cholesky in global memory)"
                        % (seed, K, M, len(win["obs"]), win["obs"].nbytes / 1e6, k_opt, 6 * k_opt),
               lm_iters_per_s=round(its / dt, 2), ms_per_solve=round(dt / reps * 1e3, 3), ms_per_solve_median=round(float(np.median(per)) * 1e3, 3), solves_timed=reps,
               iterations=r["iterations"],
               initial_error_px=round(r["initial_error"], 4), final_error_px=round(r["final_error"], 4),
               kernel_ms_per_iteration={k: round(v[0] / n_it, 4) for k, v in sorted(kt.items()) if k.startswith("ba_")},
               kernel_ms_per_solve=round(kernel_ms, 3), wall_over_kernels=round(dt / reps * 1e3 / kernel_ms, 3) if kernel_ms > 0 else None,
               obs_32_bytes=dict(lm_iters_per_s=round(its64 / dt64, 2), ms_per_solve=round(dt64 / reps * 1e3, 3), MB_up_per_solve=round(win64["obs"].nbytes / 1e6, 1),
                                 same_result_bit_for_bit=bool(np.array_equal(r["poses_wc"], r64["poses_wc"]) and np.array_equal(r["points"], r64["points"]))),
               algorithmic_GFLOP_per_iteration=round(ba_algorithmic_flop(k_opt, M, len(win["obs"])) / 1e9, 3),   # SURVEY §8(d)'s formula at this window's K, M, N
               schur=dict(executed_GFLOP_per_launch=round(flop / 1e9, 3), fused_launch_ms=round(fused_ms, 4),
                          executed_TFLOPs_fused_launch=round(flop / (fused_ms * 1e-3) / 1e12, 2) if fused_ms > 0 else None,
                          mfma_frac_fused_launch=round(flop / (fused_ms * 1e-3) / 78.6e12, 4) if fused_ms > 0 else None,
                          note="one window: the keyframe partials share the Schur launch, so this fraction is a lower bound for the MFMA part"))
    # the Schur product alone: 4 such windows through the batch call, where it is its own launch
    wins = [win64] + [P.synth.ba_window(seed + 1 + i, K, M, P.BA_OBS) for i in range(3)]
    h.ba_solve_visual_batch(cam, cfg, wins)
    h.set_profiling(True)
    rb = h.ba_solve_visual_batch(cam, cfg, wins)
    kb = h.kernel_times()
    h.set_profiling(False)
    sch = kb.get("ba_schur_kernel", (0.0, 1))
    sch_ms = sch[0] / max(sch[1], 1)
    fl4 = sum(schur_executed_flop(len(w["poses_cw"]), M) for w in wins)
    res["schur"].update(batch_of_4=dict(schur_launch_ms=round(sch_ms, 4), executed_TFLOPs=round(fl4 / (sch_ms * 1e-3) / 1e12, 2) if sch_ms > 0 else None,
                                        mfma_frac=round(fl4 / (sch_ms * 1e-3) / 78.6e12, 4) if sch_ms > 0 else None,
                                        kernel_ms_per_iteration={k: round(v[0] / max(v[1], 1), 4) for k, v in sorted(kb.items()) if k.startswith("ba_")},
                                        iterations=[x["iterations"] for x in rb]))
    return res


def bench_ba_cpu(P):
    """The oracle's two LM solvers timed on the host: ba_solve_dense = the reference's literal formulation (dense J, J^T J,
    LU of all 6K+3M unknowns, local_ba_lm.rs:1012-1056) at K=8 / M=400 where it fits; ba_solve_schur = the structured variant at
    configs[2] and configs[4] sizes.  One thread, and all host threads with one independent window per thread.  A bounded
    sample (about 10-25 s), kind "port"."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    flags = "-O2 (portable build)"
    try:
        O.use_native_build()
        flags = "-O3 -march=native"
    except Exception:
        O.lib()
    ocam = O.Camera(**P.synth.EUROC_CAMERA)
    out = dict(kind="port", threads=1, cores=1, unit="LM iterations/s", build="g++ " + flags)
    for key, fn, seed, K, M, reps in (("dense_K8_M400", O.ba_solve_dense, 42, 8, 400, 1), ("schur_K20_M2000", O.ba_solve_schur, 42, 20, 2000, 6),
                                      ("schur_K50_M8000", O.ba_solve_schur, 43, 50, 8000, 1)):
        w = P.synth.ba_window(seed, K, M, P.BA_OBS)
        t0 = time.perf_counter()
        its = 0
        for _ in range(reps):
            its += fn(ocam, O.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])["iterations"]
        dt = time.perf_counter() - t0
        out[key] = dict(lm_iters_per_s=round(its / dt, 3), ms_per_iteration=round(dt / max(its, 1) * 1e3, 3), observations=len(w["obs"]),
                        unknowns=6 * len(w["poses_cw"]) + 3 * M, solves=reps)
    # all host threads: independent windows, one per thread at a time (ctypes releases the GIL inside the oracle's solver)
    nthr = os.cpu_count() or 1
    wins = [P.synth.ba_window(200 + i, 20, 2000, P.BA_OBS) for i in range(min(nthr, 16))]

    def one(i):
        w = wins[i % len(wins)]
        return O.ba_solve_schur(ocam, O.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])["iterations"]

    n_solves = 3 * nthr
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=nthr) as ex:
        its = sum(ex.map(one, range(n_solves)))
    dt = time.perf_counter() - t0
    out["schur_K20_M2000_all_threads"] = dict(lm_iters_per_s=round(its / dt, 3), threads=nthr, cores=nthr, solves=n_solves,
                                              note="independent windows, one per host thread (BASELINE.md §3: window-level parallelism)")
    out["sample"] = ("oracle.ba_solve_dense (the reference's literal dense LM, local_ba_lm.rs:1012-1056) on synth_ba(42, 8, 400): 1 solve; "
                     "oracle.ba_solve_schur (structured) on synth_ba(42, 20, 2000): 6 solves on one thread and %d solves over %d threads, "
                     "and synth_ba(43, 50, 8000): 1 solve; 10 LM iterations each" % (n_solves, nthr))
    return out


def ba_algorithmic_flop(K_opt, M, N):
    """F_iter = N*370 + N*100 + M*60 + 2*(6K)^2*(3M) + (6K)^3/3 + N*36 (structured local BA, per LM iteration)."""
    n = 6 * K_opt
    return N * 370.0 + N * 100.0 + M * 60.0 + 2.0 * n * n * 3.0 * M + n ** 3 / 3.0 + N * 36.0


def bench_ba_c_abi(P, wins, reps=8):
    """orbx_ba_solve_visual_batch timed from a compiled caller (tests/cpp/ba_batch_driver.cpp, built here with g++): the C ABI itself,
    no Python mirror in the process — with the observations in one orbx_host_alloc buffer and in pageable memory."""
    import subprocess
    import tempfile
    libdir = os.path.join(ROOT, "orb-slam3-rust_amd")
    out = {}
    # the children must not inherit a profiler's preload (bench.py itself may run under rocprofv3: its tool library initialises the GPU in
    # every process it is loaded into, and the compiled driver is a separate GPU process that the profile is not about)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTX"))}
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "ba_batch_driver")
        subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "ba_batch_driver.cpp"),
                        "-o", exe, "-L", libdir, "-lorbx_hip", "-Wl,-rpath," + libdir], check=True, capture_output=True, env=env)
        P.synth.write_ba_batch_file(os.path.join(tmp, "batch.bin"), wins, P.BA_OBS)
        for mode in ("pinned32", "pinned", "pageable"):
            r = subprocess.run([exe, os.path.join(tmp, "batch.bin"), os.path.join(tmp, "out.bin"), str(reps), mode], check=True, capture_output=True, text=True, timeout=300, env=env)
            line = json.loads(r.stdout.strip().splitlines()[-1])
            its = sum(x["iterations"] for x in P.synth.read_ba_batch_results(os.path.join(tmp, "out.bin"), wins))
            out[mode] = dict(ms_per_call_median=line["ms_per_call_median"], ms_per_call_min=line["ms_per_call_min"],
                             lm_iters_per_s=round(its / (line["ms_per_call_median"] * 1e-3), 1), obs_memory=line["obs_memory"])
    out["note"] = "tests/cpp/ba_batch_driver.cpp: the call as a compiled host makes it (median of %d calls after one untimed call)" % reps
    return out


def ba_transport_text(world, native, comm, rehearse):
    """What carried the all-reduces of the point-partitioned BA solve, for the bench line (`local_ba.transport`, and appended to
    `config.parallelism`): with the library's own communicator the rank count is the communicator's own answer (ncclCommCount)."""
    if world == 1:
        return "one GPU, no collective"
    if native and isinstance(comm, tuple):
        return "native RCCL: communicator of %s ranks (ncclCommCount), this is rank %s" % (comm[0], comm[1])
    if native:
        return "native RCCL (ncclCommCount unavailable: %s)" % (comm,)
    return "all-reduce hook over torch.distributed (%s)" % ("gloo, REHEARSAL" if rehearse else "nccl = RCCL")


def bench_ba(P, h, cam, rank=0, world=1, dev=None):
    import torch
    """configs[2]: local BA, 20 keyframes / 2000 map points, LM iterations per second.  One GPU: the
    whole window on this GPU.  N GPUs (configs[3]): the SAME window with its map points partitioned
    over the ranks and the reduced normal equations all-reduced over RCCL every iteration."""
    win = P.synth.ba_window(42, 20, 2000, P.BA_OBS)
    if world == 1:
        win = P.Handle.pack_ba_windows([win])[0]       # the caller's observation storage is page-locked (orbx.h): the copy engine reads it where it lies
    cfg = P.LocalBAConfigLM()
    hook = None
    native = False
    if world > 1:
        if os.environ.get("ORBX_DIST_REHEARSE") == "1":
            hook = P.dist.make_allreduce_hook(dev)       # ranks share cards: gloo through the hook (RCCL refuses two ranks per device)
        else:
            try:
                P.dist.init_native_rccl(h, rank, world)  # the library's own communicator: ncclAllReduce on its stream, no Python in the loop
                native = True
            except Exception as e:                       # (never seen; the hook over torch's communicator is the fallback, and the line says which ran)
                print("native RCCL init failed on rank %d: %r -- falling back to the torch.distributed hook" % (rank, e), file=sys.stderr)
                hook = P.dist.make_allreduce_hook(dev)
            flag = torch.tensor([1.0 if native else 0.0], device=dev)
            import torch.distributed as dist
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # every rank must take the same path
            if native and flag.item() == 0.0:
                h.set_rccl_comm(None); native = False; hook = P.dist.make_allreduce_hook(dev)

    def solve():
        if world > 1:
            return P.dist.ba_solve_partitioned(h, cam, cfg, win["poses_cw"], win["fixed_cw"], win["points"], win["obs"],
                                               rank, world, hook)
        return h.ba_solve_visual(cam, cfg, win["poses_cw"], win["fixed_cw"], win["points"], win["obs"])

    r = solve()
    reps = 30                                    # (5 until round 5: 4 ms of timed solves)
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
    comm = None
    if world > 1 and native:
        try:
            comm = h.rccl_world()                        # (ranks, this rank) from ncclCommCount / ncclCommUserRank
        except Exception as e:
            comm = repr(e)
    out = dict(workload="synth_ba(seed=42, K=20, M=2000), %d observations%s" % (
                   len(win["obs"]), ", points partitioned over %d ranks + %s" % (world, "native RCCL all-reduce (ncclAllReduce issued by the library)" if native else "all-reduce hook") if world > 1 else ""),
               transport=ba_transport_text(world, native, comm, os.environ.get("ORBX_DIST_REHEARSE") == "1"),
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
        # many windows per launch (SURVEY §8d "batched problems (>= 32 windows per launch)"): orbx_ba_solve_visual_batch
        nb = 32
        # pixel coordinates at keypoint precision (f32 widened to f64), as the reference's observations are (kp.pt() is a cv::Point2f,
        # local_ba_lm.rs:870-872): the 16-byte wire format orbx_ba_obs32 then carries them exactly
        bw_pageable = [P.synth.keypoint_precision(P.synth.ba_window(200 + i, 20, 2000, P.BA_OBS)) for i in range(nb)]
        # the caller's observation storage: one page-locked buffer (orbx.h: the copy engine then reads it where it lies, one copy per half);
        # the pageable form (the library stages it through its own pinned blob) is timed beside it
        bw = P.Handle.pack_ba_windows(bw_pageable)
        batch = h.prepare_ba_batch(bw_pageable, obs32=True)   # api.BaBatch: the windows kept in the ABI's own layout (observations: orbx_ba_obs32), as a host that owns its window storage would
        batch64 = h.prepare_ba_batch(bw_pageable)             # ... and with the 32-byte observations
        rb = batch.solve(cam, cfg)
        nrep = 8
        # the rate: as a caller sees it (a batch of this size runs as two halves on two streams inside the call, see orbx.h)
        t0 = time.perf_counter()
        itb = 0
        for _ in range(nrep):
            rb = batch.solve(cam, cfg)
            itb += sum(x["iterations"] for x in rb)
        dtb = time.perf_counter() - t0
        # the per-kernel times: the same calls with the per-kernel events on (the call then keeps to one stream)
        h.set_profiling(True)
        for _ in range(nrep):
            h.ba_solve_visual_batch(cam, cfg, bw)
        kt = h.kernel_times()
        h.set_profiling(False)
        dev_ms = sum(v[0] for k, v in kt.items() if k.startswith("ba_")) / nrep
        # f64 MFMA share: executed flops of the Schur launch (upper 16x16 tiles, padded) / its measured time / 78.6 TFLOP/s
        flop = schur_executed_flop(len(bw[0]["poses_cw"]), 2000) * nb
        sch = kt.get("ba_schur_kernel", (0.0, 1))
        sch_ms = sch[0] / max(sch[1], 1)
        out["batched"] = dict(windows=nb, lm_iters_per_s=round(itb / dtb, 1), ms_per_call=round(dtb / nrep * 1e3, 3),
                              vs_single_window=round(itb / dtb / out["lm_iters_per_s"], 2),
                              device_ms_per_call=round(dev_ms, 3), lm_iters_per_s_device_only=round(itb / nrep / (dev_ms * 1e-3), 1) if dev_ms > 0 else None,
                              kernel_ms_per_iteration={k: round(v[0] / max(v[1], 1), 4) for k, v in sorted(kt.items()) if k.startswith("ba_")},
                              schur_executed_TFLOPs=round(flop / (sch_ms * 1e-3) / 1e12, 2) if sch_ms > 0 else None,
                              mfma_frac=round(flop / (sch_ms * 1e-3) / 78.6e12, 4) if sch_ms > 0 else None,
                              note="orbx_ba_solve_visual_batch on a prepared api.BaBatch (the windows kept in the ABI's layout; per call: refresh of the in/out points, the C "
                                   "call, result views): upload of the caller's observations (one pinned buffer, the 16-byte form orbx_ba_obs32: f32 pixel coordinates as the "
                                   "reference's keypoints are, widened on the device) + device-side CSR build + 10 LM iterations of all windows + "
                                   "download per call, the batch as two halves on two streams inside the call; device_ms / "
                                   "kernel_ms from a second set of calls with per-kernel events (one stream); every window equals its single-window result bit for bit "
                                   "(tests/test_ba_gpu.py)")
        out["batched"]["call_vs_device_only"] = round(out["batched"]["lm_iters_per_s"] / out["batched"]["lm_iters_per_s_device_only"], 3) if dev_ms > 0 else None
        batch64.solve(cam, cfg)
        t0 = time.perf_counter()
        it64 = 0
        for _ in range(nrep):
            it64 += sum(x["iterations"] for x in batch64.solve(cam, cfg))
        dt64 = time.perf_counter() - t0
        out["batched"]["obs_32_bytes"] = dict(lm_iters_per_s=round(it64 / dt64, 1), ms_per_call=round(dt64 / nrep * 1e3, 3),
                                              note="the same prepared batch with the observations as orbx_ba_obs (f64 coordinates): twice the upload")
        t0 = time.perf_counter()
        ita = 0
        for _ in range(nrep):
            ita += sum(x["iterations"] for x in h.ba_solve_visual_batch(cam, cfg, bw))
        dta = time.perf_counter() - t0
        out["batched"]["adhoc_mirror_call"] = dict(lm_iters_per_s=round(ita / dta, 1), ms_per_call=round(dta / nrep * 1e3, 3),
                                                   note="Handle.ba_solve_visual_batch(list of dicts): the orbx_ba_window array, the in/out points and the result dicts built per call "
                                                        "(observations in one pinned buffer all the same)")
        h.ba_solve_visual_batch(cam, cfg, bw_pageable)
        t0 = time.perf_counter()
        itp = 0
        for _ in range(nrep):
            itp += sum(x["iterations"] for x in h.ba_solve_visual_batch(cam, cfg, bw_pageable))
        dtp = time.perf_counter() - t0
        out["batched"]["pageable_observations"] = dict(lm_iters_per_s=round(itp / dtp, 1), ms_per_call=round(dtp / nrep * 1e3, 3),
                                                       note="the same call with the observations in pageable numpy arrays: staged through the handle's pinned blob by its workers")
        try:
            out["batched"]["c_abi"] = bench_ba_c_abi(P, bw_pageable)
        except Exception as e:
            out["batched"]["c_abi"] = dict(error=repr(e))
        # two such batches in flight (two handles = two HIP streams, one host thread each): the host preprocessing, upload and
        # download of one batch run under the kernels of the other
        h2 = [P.Handle(cam, 100, device=dev.index if dev is not None else 0) for _ in range(2)]
        b2 = [x.prepare_ba_batch(bw_pageable, obs32=True) for x in h2]
        for x in b2:
            x.solve(cam, cfg)
        cnt2 = [0, 0]

        def work2(i):
            for _ in range(nrep + 1):
                cnt2[i] += sum(r_["iterations"] for r_ in b2[i].solve(cam, cfg))

        th2 = [threading.Thread(target=work2, args=(i,)) for i in range(2)]
        t0 = time.perf_counter()
        for t in th2:
            t.start()
        for t in th2:
            t.join()
        dt2 = time.perf_counter() - t0
        for x in h2:
            x.close()
        out["batched"]["two_batches_in_flight"] = dict(lm_iters_per_s=round(sum(cnt2) / dt2, 1), vs_single_window=round(sum(cnt2) / dt2 / out["lm_iters_per_s"], 2),
                                                       note="2 handles x 32 windows (prepared batches, orbx_ba_obs32), one host thread each, no per-kernel events")
        # configs[4]'s BA: 50 keyframes / 8000 map points (synth_ba(seed=43, K=50, M=8000)); the reduced system (n = 294)
        # no longer fits LDS, so the factorisation is the multi-kernel form
        try:
            out["config5"] = bench_ba_config5(P, h, cam, cfg)
        except Exception as e:
            out["config5"] = dict(error=repr(e))
        try:
            out["inertial"] = bench_ba_inertial(P, h, cam)
        except Exception as e:
            out["inertial"] = dict(error=repr(e))
        # the CPU side of the BA half of the metric: the reference's literal dense-LM formulation at a size where it
        # fits, and the structured (Schur) variant at the GPU's sizes — the oracle, one thread, same run
        try:
            out["cpu_baseline"] = bench_ba_cpu(P)
        except Exception as e:
            out["cpu_baseline"] = dict(error=repr(e))
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
