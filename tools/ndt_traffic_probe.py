"""Back-to-back computeDerivatives launches at config #3 for the counter passes of tools/ndt_traffic.sh.  usage: ndt_traffic_probe.py [K=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import staticmapping_amd as sm
import bench
kv = dict(a.split("=") for a in sys.argv[1:])
K = int(kv.get("K", 1))
dev = torch.device("cuda", 0)
src, tgt, T, G = bench._submap_case(5, 500_000, 4, dev)
m = sm.NdtHip(max_source_points=len(src), max_target_points=len(tgt), pair_slots=K)
for k in range(K):
    m.set_input_source(src, slot=k); m.set_input_target(tgt, slot=k)
m.set_target_cache(True)
if K == 1:
    m.align(G)
else:
    m.align_batch(K, [G] * K)
ms, pairs = m.time_derivatives(npairs=K, launches=10)
print(f"K={K} ms_per_launch={ms:.5f} pairs_per_launch={pairs:.0f} ns={len(src)}")
m.close()
