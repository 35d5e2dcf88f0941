"""The frame-end gather of the C ABI (rt_comm_* / rt_gather_tiles_device, RCCL) on the one GPU a test box has:
a communicator of one rank exercises library loading, communicator creation and the root's own path; the multi-rank
layout is checked by emulating the other ranks' sends with copies into the slots the root would receive them in
(RCCL refuses two ranks on one device, so a real exchange needs the driver's multi-GPU run)."""
import numpy as np
import pytest
import torch

import scene_cases

pytestmark = pytest.mark.gpu


def test_rccl_gather_of_one_rank_and_the_rgb8_frame_path(rt, gpu):
    hs = scene_cases.build(rt, "ragged_random_balls_53x29_4spp")
    w, h, spp = hs.width, hs.height, hs.camera.samples_per_pixel
    ds = rt.DeviceScene(hs)
    stream = torch.cuda.current_stream().cuda_stream
    frame = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    ds.render_device(rt.render_params(seed=4), frame.data_ptr(), stream)

    comm = rt.Comm.create(rt.Comm.unique_id(), 0, 1, 0)
    assert (comm.rank, comm.size) == (0, 1)
    n = rt.out_size(w, h, rt.RT_OUT_TILES, 0, 1)
    tiles = torch.zeros(n, dtype=torch.float64, device="cuda")
    ds.render_device(rt.render_params(seed=4, out_layout=rt.RT_OUT_TILES), tiles.data_ptr(), stream)
    gathered = torch.full((n,), -1.0, dtype=torch.float64, device="cuda")
    comm.gather_tiles(w, h, 8, tiles.data_ptr(), gathered.data_ptr(), 0, stream)
    out = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    rt.tiles_to_frame_device(w, h, 1, gathered.data_ptr(), out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(out, frame)

    # the 3-bytes-per-pixel route: resolve the tile buffer on the device, gather bytes, reassemble an RGB8 frame
    tiles8 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rt.resolve_rgb8_values_device(n, spp, tiles.data_ptr(), tiles8.data_ptr(), stream)
    gathered8 = torch.zeros(n, dtype=torch.uint8, device="cuda")
    comm.gather_tiles(w, h, 1, tiles8.data_ptr(), gathered8.data_ptr(), 0, stream)
    rgb = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda")
    rt.tiles_to_frame_rgb8_device(w, h, 1, gathered8.data_ptr(), rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    want = rt.resolve_rgb8_host(w, h, spp, frame.cpu().numpy()).reshape(-1)
    assert np.array_equal(rgb.cpu().numpy(), want)
    comm.close()

    # bad arguments are errors, not crashes
    with pytest.raises(rt.RtError):
        rt.Comm.create(rt.Comm.unique_id(), 1, 1, 0)


def test_three_shards_through_the_rgb8_path(rt, gpu):
    """What rt_gather_tiles_device(elem_bytes=1) leaves on the root for N = 3: every rank's RGB8 tile buffer in its slot."""
    hs = scene_cases.build(rt, "ragged_cornell_37x37_4spp")
    w, h, spp = hs.width, hs.height, hs.camera.samples_per_pixel
    ds = rt.DeviceScene(hs)
    stream = torch.cuda.current_stream().cuda_stream
    stride = rt.out_size(w, h, rt.RT_OUT_TILES, 0, 3)
    gathered8 = torch.zeros(3 * stride, dtype=torch.uint8, device="cuda")
    for r in range(3):
        n = rt.out_size(w, h, rt.RT_OUT_TILES, r, 3)
        tiles = torch.zeros(n, dtype=torch.float64, device="cuda")
        ds.render_device(rt.render_params(seed=4, shard_index=r, shard_count=3, out_layout=rt.RT_OUT_TILES), tiles.data_ptr(), stream)
        rt.resolve_rgb8_values_device(n, spp, tiles.data_ptr(), gathered8[r * stride:].data_ptr(), stream)
    rgb = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda")
    rt.tiles_to_frame_rgb8_device(w, h, 3, gathered8.data_ptr(), rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    want = rt.resolve_rgb8_host(w, h, spp, ds.render(rt.render_params(seed=4))).reshape(-1)
    assert np.array_equal(rgb.cpu().numpy(), want)
