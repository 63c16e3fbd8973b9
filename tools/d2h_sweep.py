import sys, time, json
sys.path.insert(0, ".")
import torch, numpy as np
from geograypher_amd.meshes.meshes import _ids_to_host_int64
ids = torch.randint(-1, 1200000, (16, 3000, 4000), dtype=torch.int32, device="cuda")
ref = ids.cpu().numpy().astype(np.int64)
out = {}
for step in (1, 2, 4, 8):
    for th in (8, 16, 32, 64):
        a = _ids_to_host_int64(ids, step, th)
        assert np.array_equal(a, ref)
        t0 = time.perf_counter(); a = _ids_to_host_int64(ids, step, th); dt = time.perf_counter() - t0
        out[f"s{step}_t{th}"] = round(16 / dt, 1)
print(json.dumps(out))
