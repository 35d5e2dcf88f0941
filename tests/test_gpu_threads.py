"""Renders of one scene from several host threads on several streams ("concurrent calls on the same rt_scene
read-only OK").  The scene's lock guards only its table of per-stream scratch; a render that has to drain ITS stream before a
buffer grows, or allocate, does so under that stream's own slot — a render on another stream must not wait behind it."""
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_a_render_on_one_stream_does_not_wait_for_another_streams_drain(rt, gpu):
    hs = rt.HostScene(0, width=600, aspect=1.5, spp=500, depth=50)
    w, h = hs.width, hs.height
    ds = rt.DeviceScene(hs, sample_buffer_bytes=4 << 30)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    fa = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    fb = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    small = rt.render_params(seed=1, sample_end=4)
    ds.render_device(small, fa.data_ptr(), sa.cuda_stream)   # both streams have their scratch
    ds.render_device(small, fb.data_ptr(), sb.cuda_stream)
    torch.cuda.synchronize()
    want = fa.clone()

    about_to_drain = threading.Event()
    took = {}

    def thread_b():
        for _ in range(24):  # about half a second of work in flight on stream B
            ds.render_device(rt.render_params(seed=1), fb.data_ptr(), sb.cuda_stream)
        about_to_drain.set()
        t0 = time.perf_counter()
        # a deeper path stack than stream B's scratch holds: the library drains stream B, frees and allocates — under B's slot only
        ds.render_device(rt.render_params(seed=1, sample_end=4, max_depth=200), fb.data_ptr(), sb.cuda_stream)
        took["b"] = time.perf_counter() - t0

    def thread_a():
        about_to_drain.wait()
        time.sleep(0.05)
        t0 = time.perf_counter()
        ds.render_device(small, fa.data_ptr(), sa.cuda_stream)
        took["a"] = time.perf_counter() - t0

    tb, ta = threading.Thread(target=thread_b), threading.Thread(target=thread_a)
    tb.start(); ta.start(); tb.join(); ta.join()
    torch.cuda.synchronize()
    assert took["b"] > 0.15, took            # B really waited for its stream
    assert took["a"] < 0.25 * took["b"], took  # ... and A did not wait with it
    assert torch.equal(fa, want)              # (same frame as before the commotion)
    assert np.isfinite(fb.cpu().numpy()).all()
