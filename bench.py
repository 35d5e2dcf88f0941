#!/usr/bin/env python3
"""bench.py — Msamples/s of the render hot path on N MI355X (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1 from a bare shell: starts its own N ranks, see below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full frame of the workload: every pixel x every sample traced by the HIP kernel
(rust-tracing_amd/csrc/rt_kernel.hip) through the C ABI, the scene resident in HBM before the clock starts.
With N > 1 the frame's 8x8 tiles are dealt round-robin to the ranks (no data-path collective) and ONE gather
(RCCL) brings the tile buffers to rank 0, which reassembles the frame on its GPU — all inside the timed step.
The image is the same for every N (tests/test_gpu_parity.py, tests/test_sharding_gloo.py), so total work is
fixed: strong scaling.  The gather is the library's own (rt_gather_tiles_device: grouped ncclSend / ncclRecv, RCCL loaded
by librt_amd); `--gather torch` or an RT_ERR_COMM at communicator set-up falls back to torch.distributed.gather, and the
JSON line says which one ran (`gather`).

Started with --gpus N > 1 and no WORLD_SIZE in the environment, this script launches `torch.distributed.run` with N ranks
of itself as a CHILD process — decided from the arguments and the environment alone, before anything touches a GPU — and
exits with the child's code.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# BASELINE.json configs; [1] is the one the metric is quoted on
WORKLOADS = {
    "c1": dict(name="random-spheres 400x225, 10 spp, depth 10", scene=0, width=400, aspect=16.0 / 9.0, spp=10, depth=10),
    "c2": dict(name="random-spheres 1200x800, 500 spp, depth 50", scene=0, width=1200, aspect=1.5, spp=500, depth=50),
    "c3": dict(name="cornell box 600x600, 1000 spp, depth 50", scene=6, width=600, aspect=1.0, spp=1000, depth=50),
    "c4": dict(name="final_scene 800x800, 5000 spp, depth 40", scene=8, width=800, aspect=1.0, spp=5000, depth=40,
               earth_image="synthetic:6400x3200"),
    # BASELINE.json's 8-GPU case (also fits one GPU: the sample buffer is filled and drained 38 times)
    "c5": dict(name="final_scene 1600x1600, 10000 spp, depth 50", scene=8, width=1600, aspect=1.0, spp=10000, depth=50,
               earth_image="synthetic:6400x3200"),
}
# final_scene's image texture: the reference's own asset (assets/README.md), decoded by the host library's JPEG decoder; the
# procedural stand-in of the same size only if the file is missing
EARTH = str(ROOT / "assets" / "earth-large.jpg") if (ROOT / "assets" / "earth-large.jpg").exists() else "synthetic:6400x3200"
for _w in ("c4", "c5"):
    WORKLOADS[_w]["earth_image"] = EARTH
SCENE_SEED = 1
RENDER_SEED = 1

FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (256 CU x 64 FMA/clk x 2 x 2.4 GHz); SURVEY.md Appendix B
HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X HBM3E (MI355X_MICROARCH.md)


def algorithmic_work_per_sample(cnt):
    """Algorithmic flops and bytes of one camera path, from the instrumented kernel's event counts
    (DESIGN.md "Roofline": the per-event prices)."""
    n = max(1, cnt["samples"])
    per = {k: v / n for k, v in cnt.items()}
    flops = (24 * per["node_visits"] + 30 * per["sphere_tests"] + 16 * per["quad_tests"] + 46 * per["rays"] +
             1050 * per["noise_evals"] + 40)
    bytes_ = (64 * per["node_visits"] + 64 * per["sphere_tests"] + 128 * per["quad_tests"] + 32 * per["medium_visits"] +
              3 * per["image_lookups"])
    return flops, bytes_, per


def relaunch_under_torchrun(n_ranks):
    """--gpus N > 1 from a bare shell: one child process runs torch.distributed.run with N ranks of this script.  Nothing in
    this process has touched a GPU (no HIP call, no torch.cuda call), and the launcher is a child, not an exec."""
    import subprocess
    # --standalone: torchrun's own c10d rendezvous picks a free port itself (binding port 0 here and handing the number on would
    # leave a window in which another process can take it); --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n_ranks}", str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


SETUP_TIMEOUT_S = 120  # every blocking step of the multi-rank set-up gives up after this long: an error line, never a hang


def die(rank, what):
    """One line on stderr that names the rank and the step, then a non-zero exit (os._exit: a rank stuck inside a collective or
    an RCCL bootstrap has threads that would keep a normal exit waiting)."""
    print(f"bench.py rank {rank}: {what}", file=sys.stderr, flush=True)
    os._exit(3)


def bounded(rank, what, fn, seconds=SETUP_TIMEOUT_S):
    """fn() on a worker thread, at most `seconds`: its result, or its exception re-raised here; die() if it does not return."""
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as e:  # noqa: BLE001 (handed to the caller)
            box["error"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        die(rank, f"{what} did not return within {seconds} s (a peer that never arrived, or a refused RCCL bootstrap)")
    if "error" in box:
        raise box["error"]
    return box.get("value")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--bvh", default="reference", choices=["reference", "sah"])
    ap.add_argument("--spp", type=int, default=0, help="override spp (smoke runs only: the JSON then names the reduced config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", default="native", choices=["native", "torch"],
                    help="N > 1: the frame-end gather through the C ABI's own rt_gather_tiles_device (RCCL loaded by librt_amd; "
                         "the unique id travels over torch.distributed) or through torch.distributed.gather (RCCL behind it); "
                         "native falls back to torch when the library cannot set its communicator up (RT_ERR_COMM)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 control flow on ONE GPU (all ranks share device 0, the gather goes "
                         "through host memory); not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rt = importlib.import_module("rust-tracing_amd")
    rtdist = importlib.import_module("rust-tracing_amd.dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = 0
    n_devices = rt.amd_lib().rt_device_count()
    if n_devices <= local_rank:
        die(rank, f"no HIP device for local rank {local_rank} ({n_devices} visible); the renderer has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        from datetime import timedelta
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=SETUP_TIMEOUT_S))
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timedelta(seconds=SETUP_TIMEOUT_S))
                # the first collective is where RCCL's communicator really comes up: do it here, bounded, not inside the timed region
                probe = torch.ones(1, dtype=torch.int32, device=dev)
                bounded(rank, "the first all-reduce over RCCL", lambda: (dist.all_reduce(probe), torch.cuda.synchronize()))
                if int(probe.item()) != world:
                    die(rank, f"the first all-reduce over RCCL returned {int(probe.item())}, not the world size {world}")
        except Exception as e:  # noqa: BLE001
            die(rank, f"torch.distributed set-up failed ({args.backend}, world {world}): {type(e).__name__}: {e}")

    wl = dict(WORKLOADS[args.workload])
    wl_name = wl.pop("name")
    if args.spp > 0:
        wl["spp"] = args.spp
        wl_name += f" [REDUCED to {args.spp} spp]"
    hs = rt.HostScene(wl["scene"], scene_seed=SCENE_SEED, width=wl["width"], aspect=wl["aspect"], spp=wl["spp"],
                      depth=wl["depth"], earth_image=wl.get("earth_image"), bvh=args.bvh)
    w, h, spp = hs.width, hs.height, hs.camera.samples_per_pixel
    # One launch per frame where the frame's per-sample buffer (24 B per sample) fits 16 GiB on this rank — then "one launch" is
    # "one step", and the render kernel's average duration in a rocprofv3 trace of this command is the frame's kernel time.  A
    # bigger frame (C4, C5) keeps the library's default (2 GiB, several pipelined launches per frame; `launches_per_step` below).
    n_local_tiles = (((w + 7) // 8) * ((h + 7) // 8) + world - 1) // world
    one_launch_bytes = n_local_tiles * 64 * 24 * spp
    sample_buffer = one_launch_bytes if one_launch_bytes <= (16 << 30) else 0
    scene = rt.DeviceScene(hs, device=local_rank, sample_buffer_bytes=sample_buffer)  # scene resident in HBM before timing
    launches_per_step = 1 if sample_buffer else -(-spp // max(1, (1 << 30) // (n_local_tiles * 64 * 24)))

    stream = torch.cuda.current_stream()
    frame = torch.zeros(w * h * 3, dtype=torch.float64, device=dev)
    # render = kernels + gather + D2H.  Rank 0 brings every step's frame to a pinned host buffer inside the timed
    # region (the reference's render() ends with the Vec<Color> in host memory, src/renderer.rs:49-51).
    host_frame_buf = torch.empty(w * h * 3, dtype=torch.float64, pin_memory=True) if rank == 0 else None
    if world > 1:
        stride = rtdist.shard_stride(w, h, world)
        tiles = torch.zeros(stride, dtype=torch.float64, device=dev)
        gathered = torch.zeros(stride * world, dtype=torch.float64, device=dev) if rank == 0 else None
        params = rt.render_params(seed=RENDER_SEED, shard_index=rank, shard_count=world, out_layout=rt.RT_OUT_TILES)
    else:
        params = rt.render_params(seed=RENDER_SEED)

    # The gather of the tile buffers: the library's own exchange unless asked otherwise.  Setting the communicator up can fail
    # for reasons that have nothing to do with the renderer (no librccl to dlopen, a refused bootstrap): every rank then agrees
    # (one all-reduce of the outcome) to gather through torch.distributed instead, and the JSON line records it.
    comm = None
    gather_used = "none (one rank)" if world == 1 else "torch.distributed.gather"
    if rehearsal:
        gather_used = "host-staged torch.distributed.gather over gloo (rehearsal)"
    elif world > 1 and args.gather == "native":
        why = ""
        try:
            ids = [rt.Comm.unique_id() if rank == 0 else None]
        except rt.RtError as e:
            ids, why = [None], str(e)
        dist.broadcast_object_list(ids, src=0)
        if ids[0] is not None:
            try:  # (ncclCommInitRank blocks until every rank has arrived: bounded)
                comm = bounded(rank, "rt_comm_create (ncclCommInitRank through librt_amd)", lambda: rt.Comm.create(ids[0], rank, world, local_rank))
            except rt.RtError as e:
                why = str(e)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            gather_used = "rt_gather_tiles_device (librt_amd: grouped ncclSend / ncclRecv)"
        else:
            if comm is not None:
                comm.close()
                comm = None
            gather_used = "torch.distributed.gather (native set-up failed" + (": " + why if why else " on another rank") + ")"

    kernel_ms = []

    def step(timed):
        ev0 = ev1 = None
        if timed:
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
        if world > 1:
            scene.render_device(params, tiles.data_ptr(), stream.cuda_stream)
        else:
            scene.render_device(params, frame.data_ptr(), stream.cuda_stream)
        if timed:
            ev1.record(stream)
        if world > 1:
            if rehearsal:  # host-staged gather (gloo has no device tensors)
                stream.synchronize()
                h_tiles = tiles.cpu()
                h_all = torch.zeros(stride * world, dtype=torch.float64) if rank == 0 else None
                rtdist.gather_tiles(h_tiles, h_all, rank, world)
                if rank == 0:
                    gathered.copy_(h_all)
            elif comm is not None:
                comm.gather_tiles(w, h, 8, tiles.data_ptr(), gathered.data_ptr() if rank == 0 else 0, 0, stream.cuda_stream)
            else:
                rtdist.gather_tiles(tiles, gathered, rank, world)
            if rank == 0:
                rt.tiles_to_frame_device(w, h, world, gathered.data_ptr(), frame.data_ptr(), stream.cuda_stream)
        if rank == 0:
            host_frame_buf.copy_(frame, non_blocking=True)  # (on `stream`, behind the kernels; the closing synchronize waits for it)
        return ev0, ev1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    events = [step(True) for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in events]

    # --- outside the timed region: work counters (instrumented kernel, reduced spp), sanity, CPU baseline ---
    result = None
    if rank == 0:
        # ALGORITHMIC work per sample (DESIGN.md "Roofline"): the events of the flat left-first walk of
        # the reference's tree with the tight box test — counted by the instrumented kernel made to walk exactly that
        # (it agrees with the oracle's tight-mode counters, tests/test_gpu_parity.py).  What the shipped kernel
        # executes instead (its own trees, nearest child first, where the scene allows) is reported next to it.
        cnt_spp = min(spp, 4)
        scratch = torch.zeros(w * h * 3, dtype=torch.float64, device=dev)
        cnt_params = rt.render_params(seed=RENDER_SEED, sample_end=cnt_spp)
        executed = scene.render_device_counted(cnt_params, scratch.data_ptr(), stream.cuda_stream)
        ref_walk = rt.DeviceScene(hs, device=local_rank, walk=rt.RT_WALK_REFERENCE_ORDER)
        cnt = ref_walk.render_device_counted(cnt_params, scratch.data_ptr(), stream.cuda_stream)
        del ref_walk
        flops_ps, bytes_ps, per = algorithmic_work_per_sample(cnt)
        executed_per = {k: v / max(1, executed["samples"]) for k, v in executed.items()}
        samples_per_launch = w * h * spp / world
        k_ms = sum(kernel_ms) / len(kernel_ms)
        tflops = flops_ps * samples_per_launch / (k_ms * 1e-3) / 1e12
        host_frame = host_frame_buf.numpy()  # what the last timed step downloaded
        assert np.isfinite(host_frame).all() and host_frame.max() > 0.0, "rendered frame is empty or not finite"
        assert np.array_equal(host_frame, frame.cpu().numpy()), "the timed download differs from the device frame"

        total_samples = float(w) * h * spp * args.steps
        value = total_samples / elapsed / 1e6
        # HBM bytes per launch cannot be counted from inside this process: they come from the rocprofv3 PMC passes of
        # tools/profile_round.sh (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 correction), committed per workload in
        # profiles/hbm_traffic.json together with the tag of the build they were measured on
        traffic, traffic_source, hbm_frac = None, "no PMC pass of this workload is committed under profiles/", None
        traffic_parts, lane_util, valu_busy, pmc_source = None, None, None, None
        tp = ROOT / "profiles" / "hbm_traffic.json"
        sys.path.insert(0, str(ROOT / "tools"))
        try:
            from source_hash import source_hash
            sources_now = source_hash()
        except Exception:
            sources_now = None

        def stale(entry_hash):  # committed counters are quoted only for the kernel sources this run was built from
            return entry_hash is None or sources_now is None or entry_hash != sources_now

        if tp.exists() and world == 1 and args.spp == 0:
            try:
                entry = json.loads(tp.read_text()).get(args.workload)
                if entry and stale(entry.get("source_hash")):
                    traffic_source = (f"profiles/hbm_traffic.json holds this workload's counters for sources {entry.get('source_hash')} "
                                      f"({entry.get('tag')}); this run is built from {sources_now}: not quoted")
                    entry = None
                if entry:
                    traffic = entry.get("bytes_per_launch")  # render kernel + per-pixel summation (what the timed kernels move)
                    traffic_parts = {k: entry[k] for k in ("render_kernel_bytes", "sum_samples_kernel_bytes") if k in entry}
                    if traffic:  # measured bytes of one step / this run's kernel time of one step / 8 TB/s
                        hbm_frac = round(float(traffic) / (k_ms * 1e-3) / HBM_PEAK_BYTES_PER_S, 5)
                    traffic_source = f"profiles/hbm_traffic.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, build {entry.get('tag')}"
            except Exception:
                pass
        # the SQ counters that explain `frac` (lanes per VALU instruction, VALU issue): the newest committed PMC summary of this workload
        if world == 1 and args.spp == 0:
            pmcs = sorted((ROOT / "profiles").glob(f"r[0-9][0-9]_{args.workload}_pmc.json"))
            if pmcs:
                try:
                    sq = json.loads(pmcs[-1].read_text())
                    if stale(sq.get("source_hash")):
                        raise ValueError("counters of another build")
                    lane_util = round(float(sq["valu_lane_utilisation"]), 4)
                    if "valu_busy" in sq:
                        valu_busy = round(float(sq["valu_busy"]), 4)
                    pmc_source = f"profiles/{pmcs[-1].name}, rocprofv3 --pmc SQ passes of this command (one frame), build {sq.get('_tag', 'see file')}"
                except Exception:
                    pmc_source = f"profiles/{pmcs[-1].name} was measured on other kernel sources than this run's ({sources_now}): not quoted"
        result = {
            "metric": "Msamples/sec (pixels x spp / render seconds)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_name, "scene_seed": SCENE_SEED, "render_seed": RENDER_SEED, "bvh": args.bvh,
                       "width": w, "height": h, "spp": spp, "max_depth": hs.camera.max_depth,
                       "parallelism": (f"tiles{world}" if world > 1 else "single") + (" [gloo rehearsal on one GPU]" if rehearsal else "")},
            "gather": gather_used,
            "timed_region": "render kernels" + (" + gather + frame reassembly" if world > 1 else "") + f" + D2H of the frame to pinned host memory ({w * h * 24 / 1e6:.1f} MB per step)",
            "roofline": {
                "bound": "valu_f64", "achieved": round(tflops, 4), "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tflops / FP64_VALU_PEAK_TFLOPS, 5), "traffic": traffic, "hbm_frac": hbm_frac,
                "traffic_parts": traffic_parts, "traffic_source": traffic_source,
                "valu_lane_utilisation": lane_util, "valu_busy": valu_busy, "pmc_source": pmc_source,
                "kernel": "path_kernel (render megakernel) + sum_samples_kernel", "kernel_ms": round(k_ms, 3),
                "launches_per_step": launches_per_step,
                "sample_buffer_bytes": sample_buffer or "library default (2 GiB, two pipelined halves)",
                "algorithmic_flops_per_sample": round(flops_ps, 1), "algorithmic_bytes_per_sample": round(bytes_ps, 1),
                "events_per_sample": {k: round(v, 3) for k, v in per.items() if k != "samples"},
                "executed_events_per_sample": {k: round(v, 3) for k, v in executed_per.items() if k != "samples"},
                "walk": "own trees, nearest child first" if scene.stats()["ordered"] else "reference tree, reference order",
                "note": "bound: VALU issue under divergence (no MFMA: no dense contraction; not HBM: the scene is LDS / L2 "
                        "resident and `traffic` is a few per cent of 8 TB/s x kernel time); algorithmic_bytes are scene-record "
                        "bytes of the reference walk, listed for SURVEY 8(d), not an HBM figure",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, str(ROOT / "tests"))
            import oracle_lib  # the CPU restatement: timed here as the baseline, never part of the product path
            cores = oracle_lib.default_threads()
            # a bounded sample, about 10 s of the host's cores: the first spp of the same frame (c1: the whole workload)
            base_spp = {"c1": 10, "c2": 64, "c3": 256, "c4": 8, "c5": 2}[args.workload]
            base_spp = min(base_spp, spp)
            t1 = time.perf_counter()
            oracle_lib.render(hs, rt.render_params(seed=RENDER_SEED, sample_end=base_spp), threads=cores)
            dt = time.perf_counter() - t1
            # BASELINE.md's second CPU row: the same restatement with the tightened box test (flat / tight mode) — separates
            # what the algorithmic fix buys on a CPU from what the GPU buys
            t1 = time.perf_counter()
            oracle_lib.render(hs, rt.render_params(seed=RENDER_SEED, sample_end=base_spp), threads=cores, aabb_mode=oracle_lib.ORC_AABB_TIGHT)
            dt_tight = time.perf_counter() - t1
            result["cpu_baseline"] = {
                "value": round(w * h * base_spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                "sample": f"{w}x{h}, first {base_spp} of {spp} spp (cost is linear in spp), "
                          f"reference-faithful mode (recursive, untightened box test), {dt:.1f} s",
                "tight_box_test": {"value": round(w * h * base_spp / dt_tight / 1e6, 4), "unit": "Msamples/s", "cores": cores,
                                   "sample": f"same sample, narrowing slab test (same image bit for bit), {dt_tight:.1f} s"}}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
