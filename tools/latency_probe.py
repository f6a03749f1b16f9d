import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import frp_amd_loader
from frp_amd import native, weights
blob = weights.pack_blob(weights.make_synthetic_raw(7))
eng = native.Engine(0, max_batch=32, max_faces=10, max_h=1080, max_w=1920)
eng.load_weights(blob)
eng.gallery_set(np.random.default_rng(0).standard_normal((100000,512)).astype(np.float32))
for B in (1, 2, 4, 8, 32):
    fr = np.random.default_rng(B).integers(0,256,(B,1080,1920,3),dtype=np.uint8)
    eng.upload_frames(fr)
    for K in (1, 10):
        for _ in range(30): eng.process_resident(K, flags=1); eng.fetch_results()     # (clocks ramp up over the first calls of a small load)
        t=time.perf_counter()
        n=40
        for _ in range(n): eng.process_resident(K, flags=1); eng.fetch_results()
        dt=(time.perf_counter()-t)/n
        print(f"B={B:2d} K={K:2d}: {dt*1e3:7.3f} ms per call ({B/dt:7.1f} frames/s)")
