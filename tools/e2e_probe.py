"""bench.py's end_to_end leg alone (files -> poses through the C++ driver).  usage: e2e_probe.py [n_scans=1025]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
print(json.dumps(bench.end_to_end(torch.device("cuda", 0), n_scans=n), indent=1))
