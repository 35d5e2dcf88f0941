import importlib, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
rt = importlib.import_module("rust-tracing_amd")
import oracle_lib
for scene, w, spp, depth in ((0, 64, 4, 50), (6, 64, 4, 50), (0, 200, 8, 10)):
    hs = rt.HostScene(scene, width=w, aspect=1.5, spp=spp, depth=depth)
    p = rt.render_params(seed=1)
    t = time.time()
    got = rt.DeviceScene(hs).render(p)
    dt = time.time() - t
    want = oracle_lib.render(hs, p)
    bad = int((got.view(np.uint64) != want.view(np.uint64)).sum())
    print("scene", scene, w, spp, "render s", round(dt, 3), "differing values", bad, "of", got.size, flush=True)
