"""Child process of tests/test_gpu_gather_stub.py: RT_RCCL_LIB points at tests/rccl_stub/librccl_stub.so (a test-only stand-in
that pairs sends and receives inside one process), so N rt_comm "ranks" can share the one GPU of a test box and the N > 1 branch
of rt_gather_tiles_device runs for real: every rank renders its shard, the non-root ranks send, the root receives into the slots
the library computes, rt_tiles_to_frame_device reassembles, and the frame must equal the unsharded render bit for bit.
Prints one line per case and "ok" at the end; any failure is an exception (non-zero exit)."""
import importlib
import os
import sys
import threading
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
rt = importlib.import_module("rust-tracing_amd")
import scene_cases  # noqa: E402

assert "rccl_stub" in os.environ.get("RT_RCCL_LIB", ""), "this worker must run against the stub"


def run_case(name, n_ranks, root, threaded, seed=4):
    hs = scene_cases.build(rt, name)
    w, h, spp = hs.width, hs.height, hs.camera.samples_per_pixel
    ds = rt.DeviceScene(hs)
    stream = torch.cuda.current_stream().cuda_stream
    whole = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    ds.render_device(rt.render_params(seed=seed), whole.data_ptr(), stream)
    uid = rt.Comm.unique_id()
    comms = [rt.Comm.create(uid, r, n_ranks, 0) for r in range(n_ranks)]
    assert [(c.rank, c.size) for c in comms] == [(r, n_ranks) for r in range(n_ranks)]
    stride = rt.out_size(w, h, rt.RT_OUT_TILES, 0, n_ranks)
    shards, shards8 = [], []
    for r in range(n_ranks):
        n = rt.out_size(w, h, rt.RT_OUT_TILES, r, n_ranks)
        t = torch.zeros(max(n, 1), dtype=torch.float64, device="cuda")
        if n:
            ds.render_device(rt.render_params(seed=seed, shard_index=r, shard_count=n_ranks, out_layout=rt.RT_OUT_TILES), t.data_ptr(), stream)
        t8 = torch.zeros(max(n, 1), dtype=torch.uint8, device="cuda")
        if n:
            rt.resolve_rgb8_values_device(n, spp, t.data_ptr(), t8.data_ptr(), stream)
        shards.append(t); shards8.append(t8)
    torch.cuda.synchronize()
    for elem_bytes, parts, dtype in ((8, shards, torch.float64), (1, shards8, torch.uint8)):
        gathered = torch.full((stride * n_ranks,), 7, dtype=dtype, device="cuda")
        errors = []

        def one(r):
            try:
                comms[r].gather_tiles(w, h, elem_bytes, parts[r].data_ptr(), gathered.data_ptr() if r == root else 0, root, stream)
            except Exception as e:  # noqa: BLE001
                errors.append((r, e))

        if threaded:  # one host thread per rank, the root first: its receives wait for the sends of the others
            ts = [threading.Thread(target=one, args=(r,)) for r in [root] + [r for r in range(n_ranks) if r != root]]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        else:  # one thread: the senders post, then the root collects
            for r in [r for r in range(n_ranks) if r != root] + [root]:
                one(r)
        assert not errors, errors
        if elem_bytes == 8:
            out = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
            rt.tiles_to_frame_device(w, h, n_ranks, gathered.data_ptr(), out.data_ptr(), stream)
            torch.cuda.synchronize()
            assert torch.equal(out, whole), f"{name}: f64 frame of {n_ranks} gathered shards differs from the whole frame"
        else:
            rgb = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda")
            rt.tiles_to_frame_rgb8_device(w, h, n_ranks, gathered.data_ptr(), rgb.data_ptr(), stream)
            torch.cuda.synchronize()
            want = rt.resolve_rgb8_host(w, h, spp, whole.cpu().numpy()).reshape(-1)
            assert np.array_equal(rgb.cpu().numpy(), want), f"{name}: RGB8 frame of {n_ranks} gathered shards differs"
    for c in comms:
        c.close()
    print(f"case {name} ranks {n_ranks} root {root} threaded {threaded}: frames identical (f64 and RGB8)", flush=True)


if sys.argv[1:] == ["mismatch"]:
    # the two sides must agree on what a shard holds: a root that expects another frame size than the sender has is an error, not a hang
    uid = rt.Comm.unique_id()
    comms = [rt.Comm.create(uid, r, 2, 0) for r in range(2)]
    a = torch.zeros(4096, dtype=torch.float64, device="cuda")
    comms[1].gather_tiles(24, 24, 8, a.data_ptr(), 0, 0, 0)
    try:
        comms[0].gather_tiles(32, 24, 8, a.data_ptr(), a.data_ptr(), 0, 0)
    except rt.RtError as e:
        print("mismatch refused:", e, flush=True)
    else:
        raise SystemExit("a gather whose sides disagree on the frame size went through")
    # ... and a root whose peer never sends gives up (RCCL_STUB_TIMEOUT_S) instead of waiting for ever
    uid = rt.Comm.unique_id()
    lonely = rt.Comm.create(uid, 0, 2, 0)
    try:
        lonely.gather_tiles(24, 24, 8, a.data_ptr(), a.data_ptr(), 0, 0)
    except rt.RtError as e:
        print("missing peer reported:", e, flush=True)
    else:
        raise SystemExit("a gather without its peer went through")
    print("ok", flush=True)
    raise SystemExit(0)

run_case("ragged_random_balls_53x29_4spp", 3, 0, False)
run_case("ragged_random_balls_53x29_4spp", 8, 0, False)
run_case("ragged_cornell_37x37_4spp", 8, 5, True)      # a root other than rank 0, one host thread per rank
run_case("ragged_cornell_37x37_4spp", 3, 2, True)
run_case("ragged_cornell_37x37_4spp", 30, 0, False)    # more ranks than tile columns: the last shards hold one tile or none
print("ok", flush=True)
