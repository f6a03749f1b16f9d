#!/usr/bin/env python3
"""Where the embedder's two kernel families cross END TO END: a process call (resident 1080p frames, forced K, IResNet-100, 100 k
gallery, results fetched) of 60 ... 160 face slots on the Winograd family (FRP_WINO_MIN_FACES=1) and on the direct family
(=100000), same process, alternating.  The per-layer probe (tools/small_m_probe.py) put the crossover near 65 faces; the whole call
puts it at ~128 - the value of FRP_WINO_MIN_FACES in csrc/frp_api.cpp.      python tools/family_crossover.py"""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader
from frp_amd import native, weights
blob = weights.pack_blob(weights.make_synthetic_raw(7))
eng = native.Engine(0, max_batch=32, max_faces=10, max_h=1080, max_w=1920)
eng.load_weights(blob)
eng.gallery_set(np.random.default_rng(0).standard_normal((100000,512)).astype(np.float32))
fr = np.random.default_rng(1).integers(0,256,(16,1080,1920,3),dtype=np.uint8)
for B,K in ((6,10),(8,8),(8,10),(10,10),(12,10),(16,8),(16,10)):
    eng.upload_frames(fr[:B])
    row=[]
    for wm in ("1","100000"):
        os.environ["FRP_WINO_MIN_FACES"]=wm
        for _ in range(15): eng.process_resident(K, flags=1); eng.fetch_results()
        t=time.perf_counter(); n=30
        for _ in range(n): eng.process_resident(K, flags=1); eng.fetch_results()
        row.append((time.perf_counter()-t)/n*1e3)
    print(f"B={B:2d} K={K:2d} ({B*K:3d} slots): winograd family {row[0]:6.3f} ms   direct family {row[1]:6.3f} ms")
